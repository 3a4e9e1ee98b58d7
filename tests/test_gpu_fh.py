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
