"""The rest of the reference's call surface on the GPU against the CPU oracle (VERDICT r1 items 5/6 of "missing"):
`batchify_rays` / `render_rays` on the reference's packed [N, 8+3+10] ray batch (rendering.py:68-195, :227-235) with
per-ray bounds, `use_fine_only` (:138), `c2w_staticcam` (:211-216), `get_rays_batch` (ray_utils.py:46-59), `lindisp=True`
end to end (:100), a far=20 frequency-embedding render (Cambridge geometry, embedding arguments up to ~1e4), the field on
the `embed.npz` positions (+-20), poses batched into one launch sequence, memory-bounded batching, NaNs in the merge."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import parity_log as P
from tests.branch import pinned_gradients, rel, tapped, three_way

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def dropin():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "nefes_amd", "dropin")
    if p not in sys.path:
        sys.path.insert(0, p)
    import models.rendering as R
    import models.nerfh_nff as M
    import models.ray_utils as RU
    return R, M, RU


def nets(Wd, C):
    from nefes_amd.field import NeRFH_NFF
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    return coarse, fine


def kwargs(M, coarse, fine, Ni=64, tat=True, lindisp=False, fine_only=False):
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=fine_only, NeRFW=True, transient_at_test=tat, netchunk=1 << 21)
    q = lambda inputs, viewdirs, ts, network_fn, typ, output_transient, test_time, store_rgb: \
        M.run_network_NeRFH_NFF(inputs, viewdirs, ts, network_fn, typ=typ, output_transient=output_transient,
                                netchunk=args.netchunk, test_time=test_time, store_rgb=store_rgb)
    return dict(network_query_fn=q, perturb=False, N_importance=Ni, N_samples=64, network_fn=coarse, network_fine=fine,
                white_bkgd=False, raw_noise_std=0., test_time=True, args=args, lindisp=lindisp)


def params(Wd, C, dtype=torch.float32):
    return O.make_field_params("coarse", Wd, C, dtype=dtype), O.make_field_params("fine", Wd, C, dtype=dtype)


def packed_batch(H, W, focal, c2w, near, far, dtype=torch.float32):
    """The [N, 21] ray batch exactly as render() assembles it (rendering.py:203-235)."""
    o, d = O.ray_bundle(H, W, focal, c2w.to(dtype))
    v = (d / torch.norm(d, dim=-1, keepdim=True)).reshape(-1, 3)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    ones = torch.ones_like(d[:, :1])
    hist = torch.full((o.shape[0], 10), 10., dtype=dtype)
    return torch.cat([o, d, near * ones, far * ones, v, hist], -1)


def maps_and_pose_gradient(tag, hip_maps, c2w_dev, tap, Wd, C, oracle_render, pose):
    """The usual end-to-end comparison: maps against the (unpinned) fp32 / float64 oracle, the pose gradient of the bench
    loss branch-pinned.  oracle_render(dtype, c2w, **kw) -> [rgb, disp, acc, extras]."""
    rgb, feat, disp, acc = hip_maps
    outs = {}
    for dt in (torch.float32, torch.float64):
        r, d_, a_, e = oracle_render(dt, pose.to(dt))
        outs[dt] = (r, e["feat_map"], d_, a_)
    for name, got, i in (("rgb", rgb, 0), ("feat", feat, 1), ("disp", disp, 2), ("acc", acc, 3)):
        three_way(tag, name, got, outs[torch.float32][i], outs[torch.float64][i])
    (gh,) = torch.autograd.grad(O.bench_loss(rgb, feat), c2w_dev)

    def oracle_run(dt, act, zf):
        c = pose.to(dt).requires_grad_()
        r, _, _, e = oracle_render(dt, c, fine_act=act, z_fine=zf)
        return {"d c2w": torch.autograd.grad(O.bench_loss(r, e["feat_map"]), c)[0]}

    pinned_gradients(tag, {"d c2w": gh}, tap, Wd, oracle_run)


@pytest.mark.parametrize("Wd,C,Ni", [(128, 128, 64), (256, 16, 128)])
def test_batchify_and_render_rays_on_the_packed_batch(Wd, C, Ni):
    R, M, _ = dropin()
    coarse, fine = nets(Wd, C)
    kw = kwargs(M, coarse, fine, Ni)
    H, W, focal = 5, 7, 6.2
    batch = packed_batch(H, W, focal, O.bench_pose(), 0., 4.)
    # per-ray bounds, as the packed batch allows (:90-93): vary near/far over the rays
    g = torch.Generator().manual_seed(11)
    batch[:, 6] = torch.rand(H * W, generator=g) * 0.3
    batch[:, 7] = 3.5 + torch.rand(H * W, generator=g)
    b_dev = batch.to(DEV).requires_grad_()
    one = R.render_rays(b_dev[:9], **kw)                      # the inner function on a slice: same rows
    with tapped() as tap:
        ret = R.batchify_rays(b_dev, 32768, **kw)
    for k in ("rgb_map", "feat_map", "disp_map", "acc_map"):
        assert torch.equal(one[k], ret[k][:9]), k
    cfg = O.RenderCfg(N_samples=64, N_importance=Ni)
    pc, pf = params(Wd, C)
    b32 = batch.clone().requires_grad_()
    r32 = O.render_rays(b32, pc, pf, cfg)
    pc64, pf64 = params(Wd, C, torch.float64)
    b64 = batch.double().requires_grad_()
    r64 = O.render_rays(b64, pc64, pf64, cfg)
    tag = f"render_rays[{Wd},{C},{Ni}]"
    for k in ("rgb_map", "feat_map", "disp_map", "acc_map"):
        three_way(tag, k, ret[k], r32[k], r64[k])
    loss = lambda r: O.bench_loss(r["rgb_map"], r["feat_map"])
    (gh,) = torch.autograd.grad(loss(ret), b_dev)
    cols = (("d rays_o", slice(0, 3)), ("d rays_d", slice(3, 6)), ("d viewdirs", slice(8, 11)))

    def oracle_run(dt, act, zf):
        b = batch.to(dt).requires_grad_()
        (g,) = torch.autograd.grad(loss(O.render_rays(b, *params(Wd, C, dt), cfg, fine_act=act, z_fine=zf)), b)
        return {name: g[:, sl] for name, sl in cols}

    pinned_gradients(tag, {name: gh[:, sl] for name, sl in cols}, tap, Wd, oracle_run)
    assert float(gh[:, 6:8].abs().max()) == 0 and float(gh[:, 11:].abs().max()) == 0    # bounds/hist carry no gradient


def test_use_fine_only():
    R, M, _ = dropin()
    coarse, fine = nets(128, 128)
    kw = dict(kwargs(M, coarse, fine, 64, fine_only=True), use_viewdirs=True, ndc=False)
    H, W, focal = 4, 6, 5.0
    c2w = O.bench_pose().to(DEV).requires_grad_()
    with tapped() as tap:
        rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
    cfg = O.RenderCfg(N_samples=64, N_importance=64, use_fine_only=True)
    maps_and_pose_gradient("use_fine_only", (rgb, ex["feat_map"], disp, acc), c2w, tap, 128, 128,
                           lambda dt, c, **k: O.render(H, W, focal, *params(128, 128, dt), cfg, c2w=c, near=0., far=4., **k),
                           O.bench_pose())


def test_c2w_staticcam():
    """rendering.py:211-216: rays of the static camera, view directions of the moving one."""
    R, M, _ = dropin()
    coarse, fine = nets(128, 128)
    kw = dict(kwargs(M, coarse, fine, 64), use_viewdirs=True, ndc=False)
    H, W, focal = 4, 5, 4.4
    cam = O.bench_pose()
    static = O.se3_exp_pose((-0.05, 0.12, 0.02), (0.0, -0.1, 0.2))
    c2w = cam.to(DEV).requires_grad_()
    with tapped() as tap:
        rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, c2w_staticcam=static.to(DEV), near=0., far=4., **kw)
    cfg = O.RenderCfg(N_samples=64, N_importance=64)

    def oracle(dt, c, fine_act=None, z_fine=None):
        batch = packed_batch(H, W, focal, static, 0., 4., dt)
        _, d_cam = O.ray_bundle(H, W, focal, c)
        v = (d_cam / torch.norm(d_cam, dim=-1, keepdim=True)).reshape(-1, 3)
        r = O.render_rays(torch.cat([batch[:, :8], v, batch[:, 11:]], -1), *params(128, 128, dt), cfg, fine_act=fine_act,
                          z_fine=z_fine)
        return [r["rgb_map"], r["disp_map"], r["acc_map"], {"feat_map": r["feat_map"]}]

    maps_and_pose_gradient("c2w_staticcam", (rgb, ex["feat_map"], disp, acc), c2w, tap, 128, 128, oracle, cam)


def test_get_rays_batch():
    _, _, RU = dropin()
    H, W, focal = 6, 9, 7.7
    poses = torch.stack([O.bench_pose(), O.se3_exp_pose((0.3, 0.1, -0.2), (1.0, -2.0, 0.5)),
                         O.se3_exp_pose((-0.6, 0.4, 0.9), (-3.0, 0.2, 4.0))])
    pd = poses.to(DEV).requires_grad_()
    o, d = RU.get_rays_batch(H, W, focal, pd)
    assert o.shape == (3, H, W, 3) and d.shape == (3, H, W, 3)
    gen = torch.Generator().manual_seed(5)
    go, gd = torch.randn(3, H, W, 3, generator=gen), torch.randn(3, H, W, 3, generator=gen)
    (gh,) = torch.autograd.grad((o * go.to(DEV)).sum() + (d * gd.to(DEV)).sum(), pd)
    p64 = poses.double().requires_grad_()
    acc = 0.
    for b in range(3):
        ro, rd = O.ray_bundle(H, W, focal, p64[b])
        assert np.array_equal(o[b].detach().cpu().numpy(), ro.detach().float().numpy())        # origins: a broadcast copy
        np.testing.assert_allclose(d[b].detach().cpu().numpy(), rd.detach().float().numpy(), rtol=3e-7, atol=1e-7)
        acc = acc + (ro * go[b].double()).sum() + (rd * gd[b].double()).sum()
    (g64,) = torch.autograd.grad(acc, p64)
    e = rel(gh, g64)
    P.record("get_rays_batch", "d c2w", e_hip=e, e_ref=None, direct=None, bound=1e-5)
    assert e < 1e-5


def test_lindisp_end_to_end():
    R, M, _ = dropin()
    coarse, fine = nets(128, 128)
    kw = dict(kwargs(M, coarse, fine, 64, lindisp=True), use_viewdirs=True, ndc=False)
    H, W, focal = 4, 6, 5.0
    c2w = O.bench_pose().to(DEV).requires_grad_()
    with tapped() as tap:
        rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0.5, far=6., **kw)
    cfg = O.RenderCfg(N_samples=64, N_importance=64, lindisp=True)
    maps_and_pose_gradient("lindisp", (rgb, ex["feat_map"], disp, acc), c2w, tap, 128, 128,
                           lambda dt, c, **k: O.render(H, W, focal, *params(128, 128, dt), cfg, c2w=c, near=0.5, far=6., **k),
                           O.bench_pose())


@pytest.mark.parametrize("Wd,C,Ni", [(256, 16, 128), (128, 128, 64)])
def test_far20_frequency_embedding_render(Wd, C, Ni):
    """Cambridge geometry without the hash grid: near=0, far=20 (data/Cambridge_world_setup/ShopFacade/world_setup.json:2-3),
    a camera several metres from the origin: positions up to ~25, i.e. sin/cos arguments x*2^9 up to ~1.3e4 -- the
    in-kernel range reduction (field_common.h sincos_turns) against torch's fp32 sin/cos and the float64 truth."""
    R, M, _ = dropin()
    coarse, fine = nets(Wd, C)
    kw = dict(kwargs(M, coarse, fine, Ni), use_viewdirs=True, ndc=False)
    H, W = 5, 8
    focal = 744. * W / 854.
    pose = O.se3_exp_pose((0.4, -0.9, 0.15), (6.0, -3.5, 9.0))
    c2w = pose.to(DEV).requires_grad_()
    with tapped() as tap:
        rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0., far=20., **kw)
    cfg = O.RenderCfg(N_samples=64, N_importance=Ni)
    maps_and_pose_gradient(f"far20[{Wd},{C}]", (rgb, ex["feat_map"], disp, acc), c2w, tap, Wd, C,
                           lambda dt, c, **k: O.render(H, W, focal, *params(Wd, C, dt), cfg, c2w=c, near=0., far=20., **k), pose)


@pytest.mark.parametrize("Wd,C", [(256, 16), (128, 128)])
def test_field_on_embed_golden_positions(golden, Wd, C):
    """The field on the positions of tests/golden/embed.npz (|x| up to 20; SURVEY hard part 4) through the
    run_network_NeRFH_NFF call surface: raw outputs and d raw / d pts against the oracle."""
    _, M, _ = dropin()
    _, fine = nets(Wd, C)
    x = T(golden("embed")["x"])                                   # [96, 3]
    pts = x.reshape(8, 12, 3)
    gen = torch.Generator().manual_seed(2)
    dirs = torch.nn.functional.normalize(torch.randn(8, 3, generator=gen), dim=-1)
    pd = pts.to(DEV).requires_grad_()
    raw = M.run_network_NeRFH_NFF(pd, dirs.to(DEV), None, fine, typ='fine', output_transient=True, test_time=True)
    g_raw = torch.randn(raw.shape, generator=gen)
    (gh,) = torch.autograd.grad((raw * g_raw.to(DEV)).sum(), pd)
    outs = {}
    for dt in (torch.float32, torch.float64):
        pf = O.make_field_params("fine", Wd, C, dtype=dt)
        p = pts.to(dt).requires_grad_()
        r = O.query_field(pf, p, dirs.to(dt), "fine", True, True)
        (gp,) = torch.autograd.grad((r * g_raw.to(dt)).sum(), p)
        outs[dt] = (r, gp)
    tag = f"field_embed_pm20[{Wd},{C}]"
    three_way(tag, "raw", raw, outs[torch.float32][0], outs[torch.float64][0])
    three_way(tag, "d pts", gh, outs[torch.float32][1], outs[torch.float64][1])


def test_render_poses_batches_into_one_launch_sequence():
    """nefes_amd.render.render_poses (f4: poses batched, rendering.py:270-273): same bits as one render() per pose, and the
    pose gradients agree; the drop-in render_path uses it."""
    R, M, _ = dropin()
    from nefes_amd import ops
    from nefes_amd.render import render_poses
    coarse, fine = nets(128, 128)
    kw = dict(kwargs(M, coarse, fine, 64), use_viewdirs=True, ndc=False)
    H, W, focal = 6, 8, 6.6
    poses = torch.stack([O.bench_pose(), O.se3_exp_pose((0.2, 0.1, -0.1), (0.3, -0.2, 0.1)),
                         O.se3_exp_pose((-0.1, 0.3, 0.2), (-0.2, 0.1, 0.4))]).to(DEV).requires_grad_()
    ops.TIMERS = {}
    rgb, disp, acc, ex = render_poses(H, W, focal, poses, near=0., far=4., **kw)
    launches = {k: len(v) for k, v in ops.TIMERS.items()}
    ops.TIMERS = None
    assert all(n == 1 for n in launches.values()), launches        # every heavy kernel once for all three poses
    assert rgb.shape == (3, H * W, 3) and ex["feat_map"].shape == (3, H * W, 128)
    (gb,) = torch.autograd.grad(O.bench_loss(rgb, ex["feat_map"]), poses)
    gs = []
    for b in range(3):
        c = poses[b].detach().clone().requires_grad_()
        r, d_, a_, e = R.render(H, W, focal, c2w=c, near=0., far=4., **kw)
        assert torch.equal(r, rgb[b]) and torch.equal(e["feat_map"], ex["feat_map"][b]) and torch.equal(d_, disp[b])
        gs.append(torch.autograd.grad((e["feat_map"] ** 2).sum() / ex["feat_map"].numel() + (r ** 2).sum() / rgb.numel(), c)[0])
    assert rel(gb, torch.stack(gs)) < 1e-6
    # the drop-in validation loop on top of it: images in pose order, PSNR printed, colour transform skipped without encode_hist
    args = types.SimpleNamespace(encode_hist=False, nerfh_nff=True)
    rgbs, disps = R.render_path(args, poses.detach(), (H, W, focal), 32768, dict(kw, near=0., far=4.),
                                gt_imgs=np.zeros((3, H, W, 3), np.float32))
    assert rgbs.shape == (3, H, W, 3) and disps.shape == (3, H, W)
    np.testing.assert_array_equal(rgbs, rgb.detach().reshape(3, H, W, 3).cpu().numpy())


def test_memory_bounded_batching_gives_the_same_maps(monkeypatch):
    """the per-launch ray count is planned once per network shape against the device memory; a forced tiny budget
    splits the batch and the maps are unchanged bit for bit; a train-mode batch that cannot be held at once is refused."""
    R, M, _ = dropin()
    import nefes_amd.render as NR
    coarse, fine = nets(128, 128)
    kw = dict(kwargs(M, coarse, fine, 64), use_viewdirs=True, ndc=False)
    H, W, focal = 40, 64, 50.0                                         # 2560 rays
    c2w = O.bench_pose().to(DEV)
    a = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
    monkeypatch.setattr(NR, "rays_per_launch", lambda *args, **k: 1024)
    b = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
    assert torch.equal(a[0], b[0]) and torch.equal(a[3]["feat_map"], b[3]["feat_map"]) and torch.equal(a[1], b[1])
    monkeypatch.undo()
    cfg = NR._cfg(kw)
    cap = NR.rays_per_launch(cfg, coarse, fine, torch.device(DEV))
    per_ray = (2 * 137 * 4 + 80 + 40) * 128
    free, _ = torch.cuda.mem_get_info()
    assert 1024 <= cap <= NR.MAX_RAYS_PER_LAUNCH and cap * per_ray <= free + torch.cuda.memory_reserved()
    assert NR.rays_per_launch(cfg, coarse, fine, torch.device(DEV)) == cap                 # planned once, not per call
    coarse.requires_grad_(True)
    monkeypatch.setattr(NR, "rays_per_launch", lambda *args, **k: 1024)
    with pytest.raises(RuntimeError, match="train-mode batch"):
        R.render(H, W, focal, c2w=c2w, near=0., far=4., **dict(kw, test_time=False, perturb=1.))


def test_no_grad_render_with_trainable_networks_is_sliced_not_refused(monkeypatch):
    """the validation renders of a training run (run_nefes.py:427,467 -> render_path under torch.no_grad(), networks still
    requiring grad) are inference: planned with the inference budget, sliced when larger than it, bit-identical to one launch."""
    R, M, _ = dropin()
    import nefes_amd.render as NR
    coarse, fine = nets(128, 128)
    coarse.requires_grad_(True)
    fine.requires_grad_(True)
    kw = dict(kwargs(M, coarse, fine, 64), use_viewdirs=True, ndc=False)
    H, W, focal = 40, 64, 50.0
    c2w = O.bench_pose().to(DEV)
    seen = []

    def plan(cfg, a, b, dev, train=False):
        seen.append(train)
        return 1024
    with torch.no_grad():
        a = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
        monkeypatch.setattr(NR, "rays_per_launch", plan)
        b = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)                      # 2560 rays > 1024: three slices
        c = R.render(H, W, focal, c2w=c2w, near=0., far=4., **dict(kw, test_time=False))
    assert seen == [False, False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[3]["feat_map"], b[3]["feat_map"]) and c[0].shape == a[0].shape
    assert not a[0].requires_grad
    # free-memory clamp of the inference plan: never above the deterministic plan, never below the floor
    monkeypatch.undo()
    cfg = NR._cfg(kw)
    cap = NR.rays_per_launch(cfg, coarse, fine, torch.device(DEV))
    assert 1024 <= NR._clamp_to_free_memory(cap, cfg, coarse, fine, torch.device(DEV)) <= cap
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (1 << 20, 1 << 38))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a: 0)
    assert NR._clamp_to_free_memory(cap, cfg, coarse, fine, torch.device(DEV)) == 1024


def test_merge_with_nans_fills_every_slot():
    """NaN depths take the all-pairs rank; torch.sort puts NaNs last and fills every slot -- so must the kernel."""
    from nefes_amd import ops
    gen = torch.Generator().manual_seed(4)
    N, Nc, Ni = 7, 16, 24
    z = torch.sort(torch.rand(N, Nc, generator=gen) * 4, -1)[0]
    w = torch.rand(N, Nc, generator=gen)
    z[2, 5] = float("nan")
    z[4, 0] = float("nan")
    z[4, 9] = float("nan")
    z_fine, z_samples = ops.sample_pdf_merge(z.to(DEV), w.to(DEV), Ni)
    want = torch.sort(torch.cat([z.to(DEV), z_samples], -1), -1)[0]
    assert torch.equal(torch.isnan(z_fine), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(z_fine, nan=-1.), torch.nan_to_num(want, nan=-1.))


def test_every_shipped_config_runs_on_the_hip_path():
    """tests/golden/configs.json: every configuration file the reference ships, parsed by the reference's own config_parser
    (tools/make_golden_configs.py).  For each distinct render-path setting: the drop-in create_nerf(args) builds the networks,
    render() with the returned render_kwargs_test produces the oracle's maps and a finite pose gradient; the training
    configurations (stage 1 / 2) also take one render with render_kwargs_train and a backward to the weights."""
    import json
    R, M, RU = dropin()
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "configs.json")))
    assert len(cfgs) >= 33
    distinct = {}
    for name, c in cfgs.items():
        distinct.setdefault(json.dumps(c, sort_keys=True), name)
    H, W, focal = 6, 8, 10.0
    for key, name in distinct.items():
        c = json.loads(key)
        args = types.SimpleNamespace(**c, basedir="/nonexistent", expname="none", ft_path=None, no_reload=True)
        kw_train, kw_test, start, grad_vars, optimizer = M.create_nerf(args)
        coarse, fine = kw_test["network_fn"], kw_test["network_fine"]
        assert (coarse.W, coarse.W_features, kw_test["N_samples"], kw_test["N_importance"]) == (c["netwidth"], 128, c["N_samples"], c["N_importance"])
        assert (grad_vars is None) == bool(c["no_grad_update"])
        near, far = (0., 4.) if c["dataset_type"].startswith("7Scenes") else (0., 20.)
        pose = O.bench_pose()
        coarse.requires_grad_(False), fine.requires_grad_(False)
        c2w = pose.to(DEV).requires_grad_()
        rgb, disp, acc, ex = R.render(H, W, focal, chunk=c["chunk"], c2w=c2w, near=near, far=far, img_idx=torch.full((1, 10), 10.),
                                      **kw_test)
        O.bench_loss(rgb, ex["feat_map"]).backward()
        assert torch.isfinite(c2w.grad).all() and float(c2w.grad.abs().max()) > 0
        cfg = O.RenderCfg()
        cfg.N_samples, cfg.N_importance, cfg.transient_at_test = c["N_samples"], c["N_importance"], c["transient_at_test"]
        pc, pf = params(c["netwidth"], 128)
        r_rgb, _, r_acc, r_ex = O.render(H, W, focal, pc, pf, cfg, c2w=pose, near=near, far=far)
        assert rel(rgb, r_rgb) < 1e-4 and rel(ex["feat_map"], r_ex["feat_map"]) < 1e-4 and rel(acc, r_acc) < 1e-4, name
        if not c["no_grad_update"]:                       # stage 1 / stage 2: the training render of run_nefes.py:42-108
            coarse.requires_grad_(True), fine.requires_grad_(True)
            torch.manual_seed(0)
            rgb_t, _, _, ex_t = R.render(H, W, focal, chunk=c["chunk"], c2w=pose.to(DEV), near=near, far=far,
                                         img_idx=torch.full((1, 10), 10.), **kw_train)
            (M.img2mse(rgb_t, torch.zeros_like(rgb_t)) + M.img2mse(ex_t["rgb0"], torch.zeros_like(rgb_t))).backward()
            g = [p.grad for n, p in fine.named_parameters() if p.grad is not None]
            assert g and all(torch.isfinite(x).all() for x in g), name
            optimizer.step()


@pytest.mark.parametrize("Nc,Ni", [(64, 128), (64, 64), (128, 128), (256, 256), (64, 37)])
def test_fused_coarse_sampler_is_bit_identical_to_the_three_launches(Nc, Ni):
    """csrc/sample_pdf.hip coarse_sample_kernel (compositing variant D + sample_pdf + sort in one launch, several rays per wave)
    against nefes_composite_fwd(COMP_SIGMA_ONLY) + nefes_sample_pdf_merge: weights, samples and merged depths bit for bit -- with
    per-ray depths and with one shared row, with the deterministic u and with per-ray jittered u, on rays that are empty, saturated
    (alpha = 1), carry a NaN, and on a ray count that fills neither the last wave nor the last workgroup."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    gen = torch.Generator().manual_seed(100 + Nc + Ni)
    N = 37
    sigma = torch.nn.functional.softplus(torch.randn(N, Nc, generator=gen) * 3) * 0.5
    sigma[1] = 0.
    sigma[2] = 3000.
    sigma[3, Nc // 2:Nc // 2 + 3] = 5000.
    sigma[4, 7] = float("nan")
    sigma[5] = 1e-12
    z_row = ops.coarse_depth_row(Nc, 0.25, 5.0, False, DEV)
    z_rays = torch.sort(z_row[None].cpu() + (torch.rand(N, Nc, generator=gen) - .5) * 0.02, -1)[0].to(DEV).contiguous()
    u_ray = torch.rand(N, Ni, generator=gen).to(DEV)
    sg = sigma.to(DEV).reshape(N, 1, Nc).contiguous()
    for z, tag in ((z_rays, "per-ray depths"), (z_row, "shared row")):
        z_full = z if z.dim() == 2 else z[None].expand(N, Nc).contiguous()
        _, _, _, acc, _, w_ref, _ = ops.composite_fwd(sg, z_full, 16, L.COMP_SIGMA_ONLY)
        for u in (None, u_ray):
            zf_ref, zs_ref = ops.sample_pdf_merge(z_full, w_ref, Ni, u=u)
            zf, zs, w = ops.coarse_sample(sg, z, Ni, u=u, want_weights=True)
            same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.), torch.nan_to_num(b, nan=-7.)) and torch.equal(torch.isnan(a), torch.isnan(b))
            assert same(w, w_ref), (tag, "weights")
            assert same(zs, zs_ref), (tag, "z_samples", (zs != zs_ref).nonzero()[:4])
            assert same(zf, zf_ref), (tag, "z_fine")
            ok = torch.ones(N, dtype=torch.bool)
            ok[4] = False
            assert bool((zf[ok.to(DEV)].diff(dim=-1) >= 0).all())


def test_render_with_the_fused_coarse_pass_equals_the_separate_launches(monkeypatch):
    """render() at test time through the two-launch coarse pass (shared depth row, fused sampler: the default) and through the four
    separate launches (NEFES_FUSED_COARSE=0): every map and the pose gradient bit for bit, at both canonical shapes."""
    R, M, _ = dropin()
    from nefes_amd import ops
    for Wd, C, Ni in ((128, 128, 64), (256, 16, 128)):
        coarse, fine = nets(Wd, C)
        kw = dict(kwargs(M, coarse, fine, Ni), use_viewdirs=True, ndc=False)
        H, W, focal = 12, 20, 16.0
        out = {}
        for fused in (True, False):
            monkeypatch.setattr(ops, "FUSED_COARSE", fused)
            monkeypatch.setattr(ops, "TIMERS", {})
            c2w = O.bench_pose().to(DEV).requires_grad_()
            rgb, disp, acc, ex = R.render(H, W, focal, c2w=c2w, near=0., far=4., **kw)
            O.bench_loss(rgb, ex["feat_map"]).backward()
            out[fused] = (rgb.detach(), disp.detach(), acc.detach(), ex["feat_map"].detach(), c2w.grad.clone())
            assert ("coarse_sample" in ops.TIMERS) == fused and ("sample_pdf_merge" in ops.TIMERS) == (not fused)
        for a, b in zip(out[True], out[False]):
            assert torch.equal(a, b)
