"""Networks on fewer embedding octaves than the kernels compute: the reference's `--reduce_embedding 0` (half: nerfh_nff.py:307-316)
and `1` (none: :317-326) and smaller `--multires` / `--multires_views` (VERDICT r3 "What's missing" 5; rounds 1-3 raised
NotImplementedError).  create_nerf sizes the networks by the embedder's out_dim (nerfh_nff.py:633-659); such a network reads a prefix of
the 63 / 27 features, so it runs on the same HIP kernels with zero weight columns for the octaves it does not have
(nefes_amd/field.py NeRFH_NFF._kernel_params).  Fixtures: tests/golden/reduced.npz, captured from the reference by
tools/make_golden_reduced.py; host-side checks of the padding: tests/test_reduced_embedding.py.
"""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import parity_log as P
from tests.test_gpu_parity import DEV, _modules, check_end_to_end, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["mode0", "mode1", "m6v2", "mode0_w256"])
def test_render_end_to_end_vs_reference(golden, tag):
    g = golden("reduced")
    t = f"e2e.{tag}"
    Wd, C, mr, mrv, mode, Nc, Ni, H, W, focal, in_xyz, in_dir = g[f"{t}.cfg"]
    coarse, fine = _modules(int(Wd), int(C), 1.0, int(in_xyz), int(in_dir))
    for typ, m in (("coarse", coarse), ("fine", fine)):          # same construction order and seed as the reference's modules
        for k, v in m.state_dict().items():
            key = f"{t}.{typ}.{k}"
            if key in g:
                v = v.cpu()
                np.testing.assert_allclose(np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])]),
                                           g[key], rtol=0, atol=0, err_msg=key)
    check_end_to_end(g, t, int(Wd), int(C), int(Nc), int(Ni), True, 1.0, int(H), int(W), float(focal), int(in_xyz), int(in_dir))


def test_create_nerf_builds_reduced_networks_on_the_hip_path():
    """The drop-in create_nerf on a shipped configuration (tests/golden/configs.json) with --reduce_embedding 0 added: networks of
    33 / 15 inputs whose render equals the float64 oracle's."""
    import json
    import os
    from tests.test_gpu_surface import dropin
    R, M, RU = dropin()
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "configs.json")))
    c = dict(next(v for v in cfgs.values() if v["dataset_type"].startswith("7Scenes") and v["no_grad_update"]))
    c["reduce_embedding"] = 0
    args = types.SimpleNamespace(**c, basedir="/nonexistent", expname="none", ft_path=None, no_reload=True)
    kw_train, kw_test, start, grad_vars, opt = M.create_nerf(args)
    coarse, fine = kw_test["network_fn"], kw_test["network_fine"]
    assert (coarse.in_channels_xyz, coarse.in_channels_dir, fine.in_channels_xyz, fine.in_channels_dir) == (33, 15, 33, 15)
    for m in (coarse, fine):
        m.requires_grad_(False)
    H, W, focal = 4, 6, 5.0
    c2w = O.bench_pose().to(DEV).requires_grad_()
    rgb, disp, acc, ex = R.render(H, W, focal, chunk=c["chunk"], c2w=c2w, near=0., far=4., img_idx=torch.full((1, 10), 10.), **kw_test)
    (gp,) = torch.autograd.grad(O.bench_loss(rgb, ex["feat_map"]), c2w)
    names = lambda m: {k: v.detach().cpu().double() for k, v in m.named_parameters() if not k.startswith(("fusion_net", "exposure_embedding"))}
    cfg = O.RenderCfg(N_samples=c["N_samples"], N_importance=c["N_importance"], transient_at_test=c["transient_at_test"], n_freq_xyz=5, n_freq_dir=2)
    c64 = O.bench_pose(torch.float64).requires_grad_()
    r64, _, _, e64 = O.render(H, W, focal, names(coarse), names(fine), cfg, c2w=c64, near=0., far=4.)
    (g64,) = torch.autograd.grad(O.bench_loss(r64, e64["feat_map"]), c64)
    e_rgb, e_feat, e_g = rel(rgb, r64), rel(ex["feat_map"], e64["feat_map"]), rel(gp, g64)
    P.record("reduced_create_nerf", "rgb / feat / d c2w vs the float64 oracle (unpinned)", e_hip=max(e_rgb, e_feat), direct=e_g, e_ref=None, bound=1e-4)
    assert e_rgb < 1e-4 and e_feat < 1e-4, (e_rgb, e_feat)
    assert e_g < 5e-3, e_g           # unpinned gradient: kink noise of a 24-ray frame (the pinned statement is the test above)


def test_train_mode_vs_reference_golden(golden):
    """One train-mode step with half the octaves: the padded columns' gradients are dropped (NeRFH_NFF.shrink_grads), every parameter's
    gradient has the parameter's shape and matches the gradient the reference itself produced."""
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render
    g = golden("reduced")
    t = "train.mode0"
    Wd, C, Nc, Ni, H, W, focal, in_xyz, in_dir = g[f"{t}.cfg"]
    Wd, C, Nc, Ni, H, W, in_xyz, in_dir = (int(v) for v in (Wd, C, Nc, Ni, H, W, in_xyz, in_dir))
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C, in_channels_xyz=in_xyz, in_channels_dir=in_dir).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True, in_channels_xyz=in_xyz, in_channels_dir=in_dir).to(DEV)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True)
    kw = dict(network_query_fn=None, perturb=0., N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    rays_o, rays_d = O.ray_bundle(H, W, float(focal), torch.from_numpy(g[f"{t}.c2w"])[:3, :4])
    rgb, disp, acc, ex = render(H, W, float(focal), rays=(rays_o.to(DEV), rays_d.to(DEV)), near=0., far=4., **kw)
    assert rel(rgb, g[f"{t}.rgb"]) < 2e-5 and rel(ex["feat_map"], g[f"{t}.feat"]) < 5e-5
    t_rgb, t_feat = torch.from_numpy(g[f"{t}.t_rgb"]).to(DEV), torch.from_numpy(g[f"{t}.t_feat"]).to(DEV)
    loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean() + ((ex["rgb0"] - t_rgb) ** 2).mean()
    assert abs(float(loss.detach()) - float(g[f"{t}.loss"])) < 1e-5 * float(g[f"{t}.loss"])
    loss.backward()
    n, worst = 0, 0.
    for k in [k for k in g if k.startswith(f"{t}.grad.")]:
        net, name = k[len(f"{t}.grad."):].split(".", 1)
        got = dict((coarse if net == "coarse" else fine).named_parameters())[name].grad
        assert got is not None and tuple(got.shape) == g[k].shape, k
        a, b = got.detach().cpu().double().reshape(-1), torch.from_numpy(g[k]).double().reshape(-1)
        if float(b.abs().max()) == 0.:
            assert float(a.abs().max()) == 0., k
            continue
        direct = float((a - b).abs().max() / b.abs().max())
        worst = max(worst, direct)
        assert direct < 1e-3, (k, direct)                              # against the reference's own fp32 gradient (both carry kink noise)
        assert float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)) > 0.9995, k
        n += 1
    P.record("reduced_train_golden", "worst parameter gradient vs the reference's fp32 gradient", direct=worst, e_hip=None, e_ref=None, bound=1e-3)
    assert n >= 19
    # a second step after the optimizer moved the weights: the device re-pack pads the same columns
    with torch.no_grad():
        for m in (coarse, fine):
            for prm in m.parameters():
                if prm.grad is not None:
                    prm.add_(prm.grad, alpha=-1e-2)
                    prm.grad = None
    rgb2, _, _, ex2 = render(H, W, float(focal), rays=(rays_o.to(DEV), rays_d.to(DEV)), near=0., far=4., **kw)
    pc = {k: v.detach().cpu().double() for k, v in coarse.named_parameters() if not k.startswith(("fusion_net", "exposure_embedding"))}
    pf = {k: v.detach().cpu().double() for k, v in fine.named_parameters()}
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, n_freq_xyz=(in_xyz - 3) // 6, n_freq_dir=(in_dir - 3) // 6)
    r64, _, _, e64 = O.render(H, W, float(focal), pc, pf, cfg, rays=(rays_o.double(), rays_d.double()), near=0., far=4.)
    assert rel(rgb2, r64) < 1e-4 and rel(ex2["feat_map"], e64["feat_map"]) < 1e-4
    assert rel(rgb2, rgb.detach()) > 1e-3       # the step did change the render
