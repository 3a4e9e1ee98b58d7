"""GPU edge cases and the rest of the reference's call surface (run with -m gpu):
single ray, ragged tiles, unsupported sizes, train-mode forward extras, white background, explicit `rays=`,
NDC, stratified jitter, the drop-in `raw2outputs_NeRFH_NFF` / `sample_pdf` / `render_path` / `create_nerf`
entry points, and re-packing after a weight update."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import branch as B

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), (b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(b).double())
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def dropin():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "nefes_amd", "dropin")
    if p not in sys.path:
        sys.path.insert(0, p)
    import models.rendering as R
    import models.nerfh_nff as M
    import models.ray_utils as RU
    return R, M, RU


def nets(Wd=128, C=128):
    from nefes_amd.field import NeRFH_NFF
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    return coarse, fine


def kwargs(coarse, fine, Ni=64, test_time=True, tat=True, perturb=0., white=False, Nc=64):
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=tat, netchunk=1 << 21)
    return dict(network_query_fn=None, perturb=perturb, N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
                use_viewdirs=True, white_bkgd=white, raw_noise_std=0., test_time=test_time, args=args, ndc=False, lindisp=False)


def oracle_params(Wd=128, C=128):
    return O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)


def test_single_ray_and_ragged_tiles():
    R, M, RU = dropin()
    coarse, fine = nets()
    pc, pf = oracle_params()
    for H, W in [(1, 1), (1, 3), (3, 5)]:          # 1, 3, 15 rays: every tile is partial
        c2w = O.bench_pose().to(DEV).requires_grad_()
        with B.tapped() as tap:
            rgb, disp, acc, ex = R.render(H, W, 2.5, c2w=c2w, near=0., far=4., **kwargs(coarse, fine))
        cfg = O.RenderCfg(N_samples=64, N_importance=64)
        r_rgb, r_disp, r_acc, r_ex = O.render(H, W, 2.5, pc, pf, cfg, c2w=O.bench_pose(), near=0., far=4.)
        assert rgb.shape == (H * W, 3) and ex["feat_map"].shape == (H * W, 128)
        assert rel(rgb, r_rgb) < 1e-4 and rel(ex["feat_map"], r_ex["feat_map"]) < 1e-4 and rel(disp, r_disp) < 1e-4
        O.bench_loss(rgb, ex["feat_map"]).backward()

        def oracle_run(dt, act, zf):
            c = O.bench_pose(dt).requires_grad_()
            r, _, _, e = O.render(H, W, 2.5, O.make_field_params("coarse", 128, 128, dtype=dt),
                                  O.make_field_params("fine", 128, 128, dtype=dt), cfg, c2w=c, near=0., far=4., fine_act=act, z_fine=zf)
            return {"d c2w": torch.autograd.grad(O.bench_loss(r, e["feat_map"]), c)[0]}

        B.pinned_gradients(f"ragged[{H}x{W}]", {"d c2w": c2w.grad}, tap, 128, oracle_run)


def test_an_empty_ray_batch_fails_loudly_as_in_the_reference():
    """rendering.py:186-195: batchify_rays over zero rays builds an empty dict and `torch.cat` / the key lookups that follow raise;
    here the C ABI refuses N <= 0 (NEFES_E_BADARG) before any launch -- an error either way, never a launch with an empty grid."""
    from nefes_amd.render import render
    coarse, fine = nets(128, 16)
    kw = kwargs(coarse, fine)
    e = torch.zeros(0, 3, device=DEV)
    with pytest.raises((RuntimeError, ValueError, IndexError)):
        render(4, 6, 5.0, rays=(e, e), near=0., far=4., **kw)


def test_unsupported_sizes_fail_loudly():
    from nefes_amd import ops, lib as L
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.composite_fwd(torch.zeros(2, 1, 600, device=DEV), torch.zeros(2, 600, device=DEV), 0, L.COMP_SIGMA_ONLY)      # S > 512
    from nefes_amd.field import NeRFH_NFF
    odd = NeRFH_NFF('coarse', W=64, f_dim=16).requires_grad_(False).to(DEV)
    with pytest.raises(RuntimeError, match="W=64.*Compiled: fp16 two-part instances"):
        odd.packed()
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.sample_pdf_merge(torch.zeros(2, 2, device=DEV), torch.zeros(2, 2, device=DEV), 4)     # Nc < 3


def test_train_mode_forward_extras_and_white_background():
    """test_time=False (frozen weights, no jitter): coarse net runs its full static head (variant C) and render() returns
    the training extras (rendering.py:160-173)."""
    R, M, RU = dropin()
    coarse, fine = nets(256, 16)
    pc, pf = oracle_params(256, 16)
    H, W = 3, 4
    for white in (False, True):
        with torch.no_grad():
            rgb, disp, acc, ex = R.render(H, W, 3.0, c2w=O.bench_pose().to(DEV), near=0., far=4.,
                                          **kwargs(coarse, fine, Ni=128, test_time=False, tat=False, white=white))
        cfg = O.RenderCfg(N_samples=64, N_importance=128, test_time=False, transient_at_test=False, white_bkgd=white)
        r_rgb, r_disp, r_acc, r_ex = O.render(H, W, 3.0, pc, pf, cfg, c2w=O.bench_pose(), near=0., far=4.)
        assert set(ex) == set(r_ex) == {"feat_map", "rgb0", "disp0", "acc0", "z_std", "transient_sigmas", "beta", "feat0"}
        assert rel(rgb, r_rgb) < 1e-4 and rel(disp, r_disp) < 1e-4 and rel(acc, r_acc) < 1e-4
        for k in r_ex:
            assert rel(ex[k], r_ex[k]) < 1e-4, k


def test_explicit_rays_and_ndc():
    R, M, RU = dropin()
    coarse, fine = nets()
    pc, pf = oracle_params()
    H, W, f = 4, 5, 4.0
    o_ref, d_ref = O.ray_bundle(H, W, f, O.bench_pose())
    o_ref = (o_ref.reshape(-1, 3) + torch.tensor([0., 0., 3.])).clone()          # keep z away from the NDC singularity
    d_ref = d_ref.reshape(-1, 3).clone()
    for ndc in (False, True):
        oh, dh = o_ref.to(DEV).requires_grad_(), d_ref.to(DEV).requires_grad_()
        kw = kwargs(coarse, fine)
        kw["ndc"] = ndc
        with B.tapped() as tap:
            rgb, disp, acc, ex = R.render(H, W, f, rays=(oh, dh), near=0., far=1. if ndc else 4., **kw)
        cfg = O.RenderCfg(N_samples=64, N_importance=64)
        r_rgb, _, _, r_ex = O.render(H, W, f, pc, pf, cfg, rays=(o_ref, d_ref), ndc=ndc, near=0., far=1. if ndc else 4.)
        assert rel(rgb, r_rgb) < 1e-4 and rel(ex["feat_map"], r_ex["feat_map"]) < 1e-4
        O.bench_loss(rgb, ex["feat_map"]).backward()

        def oracle_run(dt, act, zf):
            oc, dc = o_ref.detach().clone().to(dt).requires_grad_(), d_ref.detach().clone().to(dt).requires_grad_()
            r, _, _, e = O.render(H, W, f, O.make_field_params("coarse", 128, 128, dtype=dt),
                                  O.make_field_params("fine", 128, 128, dtype=dt), cfg, rays=(oc, dc), ndc=ndc, near=0.,
                                  far=1. if ndc else 4., fine_act=act, z_fine=zf)
            O.bench_loss(r, e["feat_map"]).backward()
            return {"d rays_o": oc.grad, "d rays_d": dc.grad}

        B.pinned_gradients(f"explicit_rays[ndc={int(ndc)}]", {"d rays_o": oh.grad, "d rays_d": dh.grad}, tap, 128, oracle_run,
                           audit="ndc_inputs_f64" if ndc else "same_inputs")


def test_stratified_jitter_path_runs_and_is_sorted():
    from nefes_amd import ops
    R, M, RU = dropin()
    coarse, fine = nets()
    torch.manual_seed(0)
    rgb, disp, acc, ex = R.render(4, 4, 3.0, c2w=O.bench_pose().to(DEV), near=0., far=4., **kwargs(coarse, fine, perturb=1.))
    assert torch.isfinite(rgb).all() and torch.isfinite(ex["feat_map"]).all()
    z = ops.coarse_depths(16, 64, 0., 4., False, torch.rand(16, 64, device=DEV))
    assert (z[:, 1:] >= z[:, :-1]).all() and z.min() >= 0 and z.max() <= 4


def test_dropin_raw2outputs_sample_pdf_render_path(golden):
    R, M, RU = dropin()
    g = golden("composite")
    C = g["g_feat"].shape[1]
    raw, z = T(g["raw"]).to(DEV), T(g["z"]).to(DEV)
    rgb, feat, disp, acc, w, depth, t_sig, beta = M.raw2outputs_NeRFH_NFF(raw, z, output_transient=True, test_time=True,
                                                                          typ="fine", transient_at_test=True)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["A.rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.cpu().numpy(), g["A.weights"], rtol=1e-5, atol=1e-7)
    assert torch.equal(t_sig, raw[..., 3 + C + 4])
    out = M.raw2outputs_NeRFH_NFF(raw[..., 3 + C:3 + C + 1], z, test_time=True, typ="coarse")
    assert out[0] is None and out[2] is None
    np.testing.assert_allclose(out[4].cpu().numpy(), g["D.weights"], rtol=1e-5, atol=1e-7)
    gs = golden("sample_pdf")
    zc, wts = T(gs["det128.z"]).to(DEV), T(gs["w"]).to(DEV)
    s = R.sample_pdf(.5 * (zc[..., 1:] + zc[..., :-1]), wts[..., 1:-1].contiguous(), 128, det=True)
    np.testing.assert_allclose(s.cpu().numpy()[:, 1:-1], gs["det128.samples"][:, 1:-1], rtol=0, atol=2e-5)
    coarse, fine = nets()
    poses = torch.stack([O.bench_pose(), O.bench_pose()]).to(DEV)
    rgbs, disps = R.render_path(None, poses, (4, 6, 3.0), 32768, dict(kwargs(coarse, fine), near=0., far=4.),
                                gt_imgs=np.zeros((2, 4, 6, 3), np.float32))
    assert rgbs.shape == (2, 4, 6, 3) and disps.shape == (2, 4, 6) and np.array_equal(rgbs[0], rgbs[1])


def test_create_nerf_checkpoint_roundtrip_and_repack(tmp_path):
    R, M, RU = dropin()
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, reduce_embedding=-1, use_viewdirs=True, netdepth=8,
                                 netwidth=128, N_importance=64, N_samples=64, netchunk=1 << 21, no_grad_update=True,
                                 basedir=str(tmp_path), expname="exp", ft_path=None, no_reload=False, perturb=1.,
                                 white_bkgd=False, raw_noise_std=0., dataset_type="7Scenes", no_ndc=True, lindisp=False,
                                 nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, lrate=5e-4)
    os.makedirs(tmp_path / "exp")
    train_kw, test_kw, start, grad_vars, opt = M.create_nerf(args)
    assert start == 0 and grad_vars is None and test_kw["test_time"] is True and test_kw["perturb"] is False
    for net in (test_kw["network_fn"], test_kw["network_fine"]):
        net.requires_grad_(False)
    c2w = O.bench_pose().to(DEV)
    rgb0, _, _, _ = R.render(3, 3, 2.0, c2w=c2w, near=0., far=4., **test_kw)
    # perturb the fine net, save a reference-format checkpoint, reload through create_nerf
    with torch.no_grad():
        test_kw["network_fine"].static_rgb[0].bias.add_(0.25)
    rgb1, _, _, _ = R.render(3, 3, 2.0, c2w=c2w, near=0., far=4., **test_kw)            # same modules, in-place update
    assert (rgb1 - rgb0).abs().max() > 0.05                                             # the packed weights were refreshed
    torch.save({"global_step": 7, "network_fn_state_dict": test_kw["network_fn"].state_dict(),
                "network_fine_state_dict": test_kw["network_fine"].state_dict()}, tmp_path / "exp" / "000007.tar")
    _, test_kw2, start2, _, _ = M.create_nerf(args)
    assert start2 == 7
    for net in (test_kw2["network_fn"], test_kw2["network_fine"]):
        net.requires_grad_(False)
    rgb2, _, _, _ = R.render(3, 3, 2.0, c2w=c2w, near=0., far=4., **test_kw2)
    assert torch.equal(rgb1, rgb2)                                                     # frozen nets: host-packed both times


def test_hashgrid_encoding_vs_oracle():
    """Row a15 (parity UNPINNED: checked against the restatement of tiny-cuda-nn's published algorithm only)."""
    from nefes_amd import ops
    from oracle import hashgrid_ref as HG
    bound = 25.0
    table = HG.make_table(0)
    grid = ops.HashGrid(bound, table=table)
    assert table.shape[0] == HG.table_entries() == 6098120
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(257, 3, generator=g) * 2 - 1) * (bound * 0.999)
    xh = x.to(DEV).requires_grad_()
    enc = grid(xh)
    xc = x.clone().double().requires_grad_()
    ref = HG.encode(xc, table.double(), bound)
    ref32 = HG.encode(x, table, bound)
    assert enc.shape == (257, 32)
    assert rel(enc, ref32) < 2e-5                      # same cells, same weights (fp32 blend order differs)
    gen = torch.randn(257, 32, generator=g)
    enc.backward(gen.to(DEV))
    ref.backward(gen.double())
    # fp32 interpolation weights at the fine levels (pos ~ 2e3) carry ~1e-4 absolute error, which the x-gradient
    # amplifies by the level scale: the fp32 restatement itself sits ~2e-2 from the f64 one; the kernel must match
    # the fp32 restatement and be no further from f64 than that
    x32 = x.clone().requires_grad_()
    HG.encode(x32, table, bound).backward(gen)
    assert rel(xh.grad, x32.grad) < 2e-4
    assert rel(xh.grad, xc.grad) < 1.5 * rel(x32.grad, xc.grad) + 1e-4
    # positions beyond +bound: corner coordinates exceed the level resolution and the dense levels' linear index wraps more
    # than once (the kernel's division-free index path hands these to the plain modulo)
    xo = torch.rand(129, 3, generator=g) * (bound * 0.6) + bound * 0.9
    assert rel(grid(xo.to(DEV)), HG.encode(xo, table, bound)) < 2e-5


def test_hashgrid_in_front_of_the_field_mlp():
    """BASELINE config 4 shape of the path: hash-grid embedding (32 features) feeding the same 8x256 MLP, forward maps and
    the pose gradient against the oracle composed the same way (hash-grid arithmetic itself: parity unpinned)."""
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render
    from oracle import hashgrid_ref as HG
    Wd, C, bound = 256, 16, 8.0
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C, in_channels_xyz=32).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, in_channels_xyz=32, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    table = HG.make_table(0) * 3e3                        # O(0.3) features so the MLP actually sees the position
    grid = ops.HashGrid(bound, table=table)
    kw = kwargs(coarse, fine, Ni=128)
    kw["xyz_encoder"] = grid
    H, W, f = 3, 4, 3.0
    c2w = O.bench_pose().to(DEV).requires_grad_()
    rgb, disp, acc, ex = render(H, W, f, c2w=c2w, near=0., far=4., **kw)
    O.bench_loss(rgb, ex["feat_map"]).backward()
    # oracle: same composition on CPU
    pc = O.make_field_params("coarse", Wd, C, in_xyz=32)
    pf = O.make_field_params("fine", Wd, C, in_xyz=32)

    def field(p, pts, v, sigma_only, dt):
        e = HG.encode(pts.reshape(-1, 3), table.to(dt), bound)
        if sigma_only:
            return O.field_forward(p, e, sigma_only=True, in_xyz=32).reshape(pts.shape[0], pts.shape[1], 1)
        ed = O.freq_encode(v[:, None].expand(pts.shape).reshape(-1, 3), 4)
        return O.field_forward(p, torch.cat([e, ed], 1), in_xyz=32).reshape(pts.shape[0], pts.shape[1], -1)

    def oracle(dt):
        c = O.bench_pose(dt).requires_grad_()
        o, d = O.ray_bundle(H, W, f, c)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        v = d / torch.norm(d, dim=-1, keepdim=True)
        near, far = torch.zeros(o.shape[0], 1, dtype=dt), torch.full((o.shape[0], 1), 4., dtype=dt)
        z = O.coarse_depths(near, far, 64, False)
        cast = lambda p: {k: t.to(dt) for k, t in p.items()}
        w0 = O.composite(field(cast(pc), o[:, None] + d[:, None] * z[..., None], v, True, dt), z, test_time=True, typ="coarse").weights
        zs = O.inverse_cdf_samples(.5 * (z[..., 1:] + z[..., :-1]), w0[..., 1:-1], 128, det=True).detach()
        zf = torch.sort(torch.cat([z, zs], -1), -1)[0]
        out = O.composite(field(cast(pf), o[:, None] + d[:, None] * zf[..., None], v, False, dt), zf, output_transient=True,
                          test_time=True, typ="fine", transient_at_test=True)
        O.bench_loss(out.rgb, out.feat).backward()
        return out, c.grad

    out32, g32 = oracle(torch.float32)
    out64, g64 = oracle(torch.float64)
    assert rel(rgb, out32.rgb) < 1e-4 and rel(ex["feat_map"], out32.feat) < 1e-4 and rel(disp, out32.disp) < 1e-4
    e_hip, e_ref = rel(c2w.grad, g64), rel(g32, g64)
    print(f"[hashgrid] d c2w: hip-vs-f64 {e_hip:.2e}  fp32-oracle-vs-f64 {e_ref:.2e}  hip-vs-fp32-oracle {rel(c2w.grad, g32):.2e}")
    assert e_hip <= max(1e-4, 3 * e_ref)


def test_pose_refinement_loop_reduces_pose_error():
    """Mode-3 style analysis-by-synthesis (DFM_pose_refine.py:290-348) on the HIP path: optimise an se(3) delta with Adam so
    that the rendered rgb+feature maps match the maps rendered at the true pose."""
    from nefes_amd.pose import LearnPose
    from nefes_amd.render import render
    coarse, fine = nets()
    kw = kwargs(coarse, fine)
    H, W, f = 12, 16, 10.0
    true = torch.eye(4)
    true[:3, :4] = O.bench_pose()
    with torch.no_grad():
        rgb_t, _, _, ex_t = render(H, W, f, c2w=true[:3, :4].to(DEV), near=0., far=4., **kw)
    start = true.clone()
    start[:3, 3] += torch.tensor([0.03, -0.02, 0.02])
    model = LearnPose(1, True, True, init_c2w=start[None].clone()).to(DEV)
    opt = torch.optim.Adam([{"params": [model.r], "lr": 2e-3}, {"params": [model.t], "lr": 5e-3}])
    losses, err0 = [], float((start[:3, 3] - true[:3, 3]).norm())
    for it in range(40):
        c2w = model(0)
        rgb, _, _, ex = render(H, W, f, c2w=c2w[:3, :4], near=0., far=4., **kw)
        loss = ((rgb - rgb_t) ** 2).mean() + ((ex["feat_map"] - ex_t["feat_map"]) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    err1 = float((model(0)[:3, 3].detach().cpu() - true[:3, 3]).norm())
    print(f"[refine] loss {losses[0]:.3e} -> {losses[-1]:.3e}; translation error {err0:.4f} -> {err1:.4f}")
    # random seed-0 weights give a low-contrast 'fog' scene (loss ~1e-8, rotation can trade against translation):
    # require that the HIP pose gradients drive the photometric+feature loss down and the pose does not drift away
    assert losses[-1] < 0.5 * losses[0] and err1 < err0


@pytest.mark.gpu
def test_pose_refiner_graph_matches_eager():
    """one captured HIP graph per refinement iteration gives the same trajectory as the eager loop."""
    import types
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.refine import PoseRefiner
    dev = torch.device("cuda")
    H, W, focal, C = 12, 16, 26.0, 16
    coarse = NeRFH_NFF('coarse', W=256, f_dim=C).requires_grad_(False).to(dev)
    fine = NeRFH_NFF('fine', W=256, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, encode_hist=True)
    kw = dict(network_query_fn=None, perturb=False, N_importance=32, N_samples=32, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
    init = torch.eye(4, device=dev)
    init[:3, :4] = O.bench_pose().to(dev)
    hist = torch.full((1, 10), 10., device=dev)
    true = init.clone()
    true[:3, 3] += torch.tensor([0.03, -0.02, 0.04], device=dev)
    with torch.no_grad():                                            # target = fused features rendered at the true pose
        from nefes_amd.render import render
        rgb, _, _, ex = render(H, W, focal, c2w=true[:3, :4], near=0., far=4., img_idx=hist, **kw)
        rgb = coarse.affine_color_transform(args, rgb, hist, 1)
        target = coarse.run_fusion_net(rgb, ex["feat_map"], H, W, 1)[2][0].clone()
    ws = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0.0, 0.0, 0.0])
    outs = []
    for graph in (False, True):
        bn = coarse.fusion_net.net[-1]
        bn.reset_running_stats()
        r = PoseRefiner(kw, args, (4 * H, 4 * W, 4 * focal), 0., 4., tinyscale=4, lr_t=0.01, world_setup=ws, graph=graph, device=dev,
                         adam_capturable=True)   # same Adam arithmetic in both arms: the loss here is noise-level, Adam amplifies
        r.refine(init, target, hist, 2)                              # second call below re-uses the captured graph
        outs.append(r.refine(init, target, hist, 4))
    (p0, l0), (p1, l1) = outs
    assert torch.isfinite(l0).all()
    assert torch.allclose(l0, l1, rtol=1e-2, atol=1e-6), (l0, l1)
    assert torch.allclose(p0, p1, rtol=0, atol=2e-3), (p0 - p1).abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((1, 16, 60, 80), (240, 320)), ((2, 3, 7, 5), (19, 23)), ((1, 4, 9, 9), (9, 9))])
def test_bicubic_upsample_matches_torch(shape, size):
    """nefes_bicubic_up_fwd/bwd against torch.nn.Upsample(size, mode='bicubic') (DFM_APR_refine.py:114) computed by
    torch on the CPU in float64; the gather backward must equal autograd's scatter backward."""
    from nefes_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(*shape, generator=g)
    go = torch.randn(shape[0], shape[1], *size, generator=g)
    xr = x.double().requires_grad_()
    yr = torch.nn.Upsample(size=size, mode='bicubic')(xr)
    yr.backward(go.double())
    xg = x.cuda().requires_grad_()
    y = ops.bicubic_upsample(xg, size)
    y.backward(go.cuda())
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=0, atol=2e-6 * float(yr.detach().abs().max()))
    assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=0, atol=2e-6 * float(xr.grad.abs().max()))


def _known_noise(shape, device="cpu", dtype=torch.float32):
    """A fixed 'draw': a function of the element index only, so the coarse [N, Nc] and the fine [N, Nc + Ni] draws of both
    implementations agree whatever their shapes' rank."""
    n = int(np.prod(tuple(shape)))
    return torch.sin(0.37 * torch.arange(n, dtype=torch.float64) + 1.0).reshape(tuple(shape)).to(dtype).to(device)


@pytest.mark.parametrize("case", ["train_mode_nerfw", "train_mode_nerfw_off", "test_time"])
def test_raw_noise_std_reaches_the_static_density_where_the_reference_draws_it(case, monkeypatch):
    """raw2outputs_NeRFH_NFF adds `randn_like(static_sigmas) * raw_noise_std` to the static density of every composite WITHOUT the
    transient head (nerfh_nff.py:66-68): the coarse pass in both modes and a fine network with NeRFW off; with the transient head the
    line is not reached (:61-64).  The draw itself is the generator's (device vs CPU streams share nothing), so both sides get the same
    known tensor in its place; maps, extras and the branch-pinned pose gradient then follow the usual rules."""
    from nefes_amd import render as RR
    R, M, RU = dropin()
    coarse, fine = nets(128, 16)
    pc, pf = oracle_params(128, 16)
    H, W, f, std = 3, 4, 3.0, 0.1
    test_time, nerfw = case == "test_time", case != "train_mode_nerfw_off"
    kw = kwargs(coarse, fine, Ni=64, test_time=test_time, tat=True)
    kw["raw_noise_std"] = std
    kw["args"].NeRFW = nerfw
    monkeypatch.setattr(RR, "_noise", lambda shape, device: _known_noise(shape, device))
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: _known_noise(t.shape, t.device, t.dtype))
    c2w = O.bench_pose().to(DEV).requires_grad_()
    with B.tapped() as tap:
        rgb, disp, acc, ex = R.render(H, W, f, c2w=c2w, near=0., far=4., **kw)
    cfg = O.RenderCfg(N_samples=64, N_importance=64, test_time=test_time, transient_at_test=True, NeRFW=nerfw, raw_noise_std=std)
    r_rgb, r_disp, r_acc, r_ex = O.render(H, W, f, pc, pf, cfg, c2w=O.bench_pose(), near=0., far=4.)
    assert rel(rgb, r_rgb) < 1e-4 and rel(disp, r_disp) < 1e-4 and rel(acc, r_acc) < 1e-4
    assert set(ex) == set(r_ex)
    for k in r_ex:
        assert rel(ex[k], r_ex[k]) < 1e-4, k
    # ... and the noise is really in there: the same call without it gives other maps
    kw0 = dict(kw, raw_noise_std=0.)
    with torch.no_grad():
        rgb_0, _, acc_0, _ = R.render(H, W, f, c2w=O.bench_pose().to(DEV), near=0., far=4., **kw0)
    assert rel(rgb, rgb_0) > 1e-5            # (through the fine samples' positions only where the fine pass has its transient head: 7e-4 here)
    O.bench_loss(rgb, ex["feat_map"]).backward()

    def oracle_run(dt, act, zf):
        c = O.bench_pose(dt).requires_grad_()
        r, _, _, e = O.render(H, W, f, O.make_field_params("coarse", 128, 16, dtype=dt), O.make_field_params("fine", 128, 16, dtype=dt),
                              cfg, c2w=c, near=0., far=4., fine_act=act, z_fine=zf)
        return {"d c2w": torch.autograd.grad(O.bench_loss(r, e["feat_map"]), c)[0]}

    if test_time:          # (train mode differentiates through the coarse pass too; its branch pattern is not tapped here)
        B.pinned_gradients(f"raw_noise[{case}]", {"d c2w": c2w.grad}, tap, 128, oracle_run)
    else:
        e = rel(c2w.grad, oracle_run(torch.float64, None, None)["d c2w"])
        assert e < 2e-2, e
