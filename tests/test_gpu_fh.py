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
