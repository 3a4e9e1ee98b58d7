"""The factored feature head (round 5; csrc/field_fwd_h3.hip FH, nefes_amd/render.py): a frozen width-128 fine network emits
g = relu(dir_encoding) (64 channels) + a channel of ones instead of its 128 feature channels, the compositor runs on those, and
W_f is applied once per ray -- exact algebra (raw2outputs is linear in the head's outputs with weights that do not depend on them,
script/models/nerfh_nff.py:119-125; the head has no activation, :487-490), different rounding.  Against the plain kernels and
against the oracle, at the shape every shipped configuration runs (8 x 128, C = 128, 64 + 64 samples)."""
import pytest
import torch

from oracle import ref_cpu as O
from tests import parity_log as P
from tests.branch import rel, tapped
from tests.test_gpu_surface import dropin, kwargs, maps_and_pose_gradient, nets, params

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tat,Ni", [(True, 64), (False, 64), (True, 192)])
def test_factored_head_equals_the_plain_kernels_and_the_oracle(tat, Ni):
    from nefes_amd import ops
    R, M, _ = dropin()
    Wd, C = 128, 128
    coarse, fine = nets(Wd, C)
    kw = dict(kwargs(M, coarse, fine, Ni, tat=tat), use_viewdirs=True, ndc=False)
    H, W, focal = 6, 8, 7.0
    pose = O.bench_pose()
    assert fine.factored_head_ok()
    outs = {}
    for fh in (True, False):
        ops.FACTORED_HEAD = fh
        ops.TIMERS = timers = {}
        try:
            c2w = pose.to(DEV).requires_grad_()
            with tapped() as tap:
                rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
            (g,) = torch.autograd.grad(O.bench_loss(rgb, ex["feat_map"]), c2w, retain_graph=True)
            outs[fh] = (rgb, ex["feat_map"], disp, acc, g, c2w, tap, set(timers))
        finally:
            ops.FACTORED_HEAD = True
            ops.TIMERS = None
    assert {"field_fwd[full,h3,fh]", "field_bwd[h3,fh]"} <= outs[True][7] and not any("fh" in k for k in outs[False][7])
    tag = f"factored_head[tat={int(tat)},Ni={Ni}]"
    for name, i, tol in (("rgb", 0, 2e-6), ("feat", 1, 2e-6), ("disp", 2, 2e-6), ("acc", 3, 2e-6), ("d c2w", 4, 5e-5)):
        e = rel(outs[True][i], outs[False][i])
        P.record(tag, f"{name}: factored head vs plain kernels", direct=e, bound=tol)
        assert e < tol, (name, e)
    # against the oracle (maps three-way, pose gradient on the kernels' own branches), like every other render test
    rgb, feat, disp, acc, _, c2w, tap, _ = outs[True]
    cfg = O.RenderCfg(N_samples=64, N_importance=Ni)
    cfg.transient_at_test = tat
    maps_and_pose_gradient(tag, (rgb, feat, disp, acc), c2w, tap, Wd, C,
                           lambda dt, c, **k: O.render(H, W, focal, *params(Wd, C, dt), cfg, c2w=c, near=0., far=4., **k), pose)


def test_factored_head_is_not_used_where_it_does_not_apply():
    """Trainable weights, the headline network (3 + 16 channels against 128), width 256 with 128 channels (131 against 129): plain kernels."""
    from nefes_amd.field import NeRFH_NFF
    mk = lambda Wd, C: NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).to(DEV)
    assert mk(128, 128).requires_grad_(False).factored_head_ok()
    assert not mk(128, 128).factored_head_ok()                       # trainable
    assert not mk(256, 16).requires_grad_(False).factored_head_ok()
    assert not mk(256, 128).requires_grad_(False).factored_head_ok()
    assert not mk(128, 16).requires_grad_(False).factored_head_ok()
    assert mk(128, 96).requires_grad_(False).factored_head_ok()


def test_single_node_fine_pass_equals_the_three_node_chain():
    """ops.RenderFineFH (the compositor leaves the static weight where the g channels' gradient would go, the field backward forms
    w_s g_gmap[ray] itself) against the chain FieldFromRaysFH -> Composite -> FeatHead (64 rows of products written and read): same
    maps bit for bit, same ray gradients to rounding."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    _, fine = nets(128, 128)
    pk, w_f, w_f_t, b_f = fine.packed_fh()
    g = torch.Generator().manual_seed(9)
    N, S = 53, 128
    o = (torch.randn(N, 3, generator=g) * 0.3).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].to(DEV)
    G_rgb, G_feat = torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, 128, generator=g).to(DEV)
    flags = L.COMP_TRANSIENT
    res = []
    for single in (True, False):
        oo, dd, vv = (t.clone().requires_grad_() for t in (o, d, d))
        if single:
            rgb, feat, disp, acc = ops.RenderFineFH.apply(oo, dd, vv, z, pk, w_f, w_f_t, b_f, flags, 0.1)
        else:
            raw = ops.FieldFromRaysFH.apply(oo, dd, vv, z, pk)
            rgb, gmap, disp, acc, _, _, _ = ops.Composite.apply(raw, z, 65, flags, 0.1)
            feat = ops.FeatHead.apply(gmap, w_f, w_f_t, b_f)
        ((rgb * G_rgb).sum() + (feat * G_feat).sum()).backward()
        res.append((rgb.detach(), feat.detach(), disp.detach(), acc.detach(), oo.grad, dd.grad, vv.grad))
    for i in range(4):
        assert torch.equal(res[0][i], res[1][i])
    for i, name in ((4, "d rays_o"), (5, "d rays_d"), (6, "d viewdirs")):
        e = rel(res[0][i], res[1][i])
        P.record("factored_head_single_node", f"{name}: one node vs the three-node chain", direct=e, bound=2e-6)
        assert e < 2e-6, (name, e)


def test_feature_head_folded_into_fusion_nets_first_convolution():
    """render(..., feat_as_gmap=True) hands over the factored head's per-ray input (sum_s w_s g_s, sum_s w_s) in the features' place, and
    FusionNet.forward_prepared_gmap runs its first convolution on weights composed with the head's (the refinement loop's path since
    round 5): the head applied to that input is the feature map of the ordinary render, and the fused features and the pose gradient
    through FusionNet are those of the ordinary chain (render -> per-ray head -> fusion_input -> four convolutions -> BatchNorm)."""
    from nefes_amd import ops
    R, M, _ = dropin()
    Wd, C = 128, 128
    coarse, fine = nets(Wd, C)
    coarse, fine = coarse.requires_grad_(False), fine.requires_grad_(False)
    kw = dict(kwargs(M, coarse, fine, 64, tat=True), use_viewdirs=True, ndc=False)
    H, W, focal = 6, 8, 7.0
    fnet = coarse.fusion_net
    assert not fnet.fusion_residule and fnet._use_hip(torch.zeros(1, device=DEV))
    _, w_f, _, b_f = fine.packed_fh()
    gen = torch.Generator().manual_seed(4)
    G = torch.randn(1, C, H, W, generator=gen).to(DEV)
    outs = {}
    for gm in (False, True):
        c2w = O.bench_pose().to(DEV).requires_grad_()
        rgb, _, _, ex = R.render(H, W, focal, c2w=c2w, near=0., far=4., **dict(kw, feat_as_gmap=gm))
        assert bool(ex.get("feat_is_gmap", False)) == gm
        feat = ex["feat_map"]
        x = ops.fusion_input(rgb, feat, None, 1, H, W, fnet.mean, fnet.std)
        fused = fnet.forward_prepared_gmap(x, w_f, b_f) if gm else fnet.forward_prepared(x)
        (g,) = torch.autograd.grad((fused * G).sum(), c2w)
        outs[gm] = (feat.detach(), fused.detach(), g)
    gmap = outs[True][0]
    assert gmap.shape == (H * W, Wd // 2 + 1) and outs[False][0].shape == (H * W, C)
    head = gmap[:, :-1].double() @ w_f.double().t() + gmap[:, -1:].double() * b_f.double()
    e_feat, e_fused, e_g = rel(head, outs[False][0]), rel(outs[True][1], outs[False][1]), rel(outs[True][2], outs[False][2])
    P.record("gmap_conv0", "feature head applied to the handed-over input vs feat_map", direct=e_feat, bound=2e-6)
    P.record("gmap_conv0", "fused features: head folded into conv0 vs per-ray head", direct=e_fused, bound=1e-5)
    P.record("gmap_conv0", "d c2w through FusionNet: head folded into conv0 vs per-ray head", direct=e_g, bound=5e-5)
    assert e_feat < 2e-6 and e_fused < 1e-5 and e_g < 5e-5, (e_feat, e_fused, e_g)
    # where the factored head does not apply (here: switched off) the flag is ignored and says so
    ops.FACTORED_HEAD = False
    try:
        _, _, _, ex = R.render(H, W, focal, c2w=O.bench_pose().to(DEV), near=0., far=4., **dict(kw, feat_as_gmap=True))
        assert "feat_is_gmap" not in ex and ex["feat_map"].shape == (H * W, C)
    finally:
        ops.FACTORED_HEAD = True


@pytest.mark.parametrize("Wd,C,tat", [(128, 128, True), (128, 128, False), (256, 16, True)])
def test_white_background_at_test_time_maps_and_pose_gradient(Wd, C, tat):
    """white_bkgd=True (script/models/nerfh_nff.py:126-127: rgb += 1 - acc) through the test-time fine pass -- at the shipped shape that
    is ops.RenderFineFH with COMP_WHITE_BKGD (nefes_amd/render.py), which round 5 left tested under no_grad, at (256, 16), test_time=False
    only: maps three-way against the oracle and the pose gradient on the kernels' own branches, both transient_at_test values at the
    factored-head shape, and the headline shape with gradients."""
    from nefes_amd import ops
    R, M, _ = dropin()
    coarse, fine = nets(Wd, C)
    Ni = 64 if Wd == 128 else 128
    kw = dict(kwargs(M, coarse, fine, Ni, tat=tat), use_viewdirs=True, ndc=False, white_bkgd=True)
    H, W, focal = 6, 8, 7.0
    pose = O.bench_pose()
    ops.TIMERS = timers = {}
    try:
        c2w = pose.to(DEV).requires_grad_()
        with tapped() as tap:
            rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
    finally:
        ops.TIMERS = None
    assert ("field_fwd[full,h3,fh]" in timers) == (Wd == 128), sorted(timers)          # the factored head ran where it applies
    # the white background is really in the map: without it the colours differ by 1 - acc
    with torch.no_grad():
        rgb_b = R.render(H, W, focal, c2w=pose.to(DEV), near=0., far=4., **dict(kw, white_bkgd=False))[0]
    assert rel(rgb - rgb_b, (1. - acc)[:, None].expand_as(rgb)) < 1e-5
    cfg = O.RenderCfg(N_samples=64, N_importance=Ni, white_bkgd=True)
    cfg.transient_at_test = tat
    maps_and_pose_gradient(f"white_bkgd[{Wd},{C},tat={int(tat)}]", (rgb, ex["feat_map"], disp, acc), c2w, tap, Wd, C,
                           lambda dt, c, **k: O.render(H, W, focal, *params(Wd, C, dt), cfg, c2w=c, near=0., far=4., **k), pose)
