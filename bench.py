#!/usr/bin/env python3
"""Benchmark of the NeFeS render-and-refine hot path on MI355X (BASELINE.json metric).

One "step" = one full render() forward + backward to the 3x4 camera pose over one synthetic frame:
640x480 rays, 64 coarse + 128 importance samples, 8x256 MLP with a 16-channel feature head, random
(seed-0) weights, loss = mean(feat^2) + mean(rgb^2)  (SURVEY.md §8d).  With N GPUs the frame's rows
are sharded across ranks (strong scaling) and the only collective is the 48-byte pose-gradient
all-reduce.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic MACs per sample (frozen weights => backward = dX only = forward MACs)
def macs_sigma(W, in_xyz=63):
    return in_xyz * W + 3 * W * W + (W + in_xyz) * W + 3 * W * W + W


def macs_full(W, C, in_xyz=63):
    h = W // 2
    return macs_sigma(W, in_xyz) + W * W + (W + 27) * h + h * (3 + C) + (W + 27) * h + 2 * h * h + 5 * h


# workloads: "metric" is the BASELINE.json headline (configs[1]); the others are secondary lines, never the default
WORKLOADS = {
    "metric": dict(H=480, W=640, focal=525.505, near=0., far=4., Wd=256, C=16, Nc=64, Ni=128, hashgrid=False,
                   name="BASELINE configs[1]: 7-Scenes-stairs geometry, 64+128 samples, 8x256 MLP + 16-ch feature head"),
    "metric128": dict(H=480, W=640, focal=525.505, near=0., far=4., Wd=256, C=128, Nc=64, Ni=128, hashgrid=False,
                      name="the BASELINE frame with the network the reference itself builds at --netwidth 256: FEATURE_DIM = 128 is a "
                           "module constant (nerfh_nff.py:21,427), so 8x256 MLP + 128-ch feature head, 64+128 samples"),
    "cam": dict(H=480, W=854, focal=744., near=0., far=20., Wd=256, C=16, Nc=64, Ni=128, hashgrid=True,
                name="BASELINE configs[3]: Cambridge ShopFacade geometry 854x480, hash-grid (L=16,F=2,T=2^19, bound 25) in "
                     "front of the 8x256 MLP + 16-ch feature head (hash-grid arithmetic: parity unpinned)"),
    "ref": dict(H=60, W=80, focal=262.75 / 4, near=0., far=4., Wd=128, C=128, Nc=64, Ni=64, hashgrid=False,
                name="refinement-loop frame the reference actually renders (DFM_APR_refine.py:107): 80x60, 64+64 samples, "
                     "8x128 MLP + 128-ch feature head (reference defaults)"),
}


PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0    # same rate: v_mfma_f32_32x32x16_f16 = 32 cycles per 32x32x16 like the bf16 form
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def bench_pose():
    """SURVEY.md §8d pose: c2w = [Exp(r) | t], r = (0.10, -0.20, 0.05), t = (0.10, 0.20, 0.30) (nefes_amd.pose, fp64 -> fp32)."""
    from nefes_amd.pose import make_c2w
    return make_c2w(torch.tensor([0.10, -0.20, 0.05], dtype=torch.float64),
                    torch.tensor([0.10, 0.20, 0.30], dtype=torch.float64))[:3, :4].float()


def cpu_baseline(Wd, C, Nc, Ni, n_rows, W, focal):
    """The CPU oracle (same torch op sequence as the reference) on a bounded slice of the same workload."""
    from oracle import ref_cpu as O
    cores = host_cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni)

    def once():
        c2w = O.bench_pose().requires_grad_()
        # rows [0, n_rows) of the 480x640 frame: same geometry as the GPU workload
        rays_o, rays_d = O.ray_bundle(480, W, focal, c2w)
        rgb, _, _, ex = O.render(480, W, focal, pc, pf, cfg, rays=(rays_o[:n_rows], rays_d[:n_rows]), near=0., far=4.)
        O.bench_loss(rgb, ex["feat_map"]).backward()
        return c2w.grad

    # pick the thread count the host actually runs this op mix best at (using every hardware thread of a
    # 256-thread host on 256-wide GEMMs is many times slower than a moderate count), on a 1/4-size calibration run
    full_rows, n_rows = n_rows, max(1, n_rows // 4)
    best = (float("inf"), cores)
    for th in sorted({min(cores, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        once()
        dt = time.perf_counter() - t0
        best = min(best, (dt, th))
        if dt > 20.0:
            break
    cores = best[1]
    torch.set_num_threads(cores)
    n_rows = full_rows
    t0 = time.perf_counter()
    once()
    dt = time.perf_counter() - t0
    n = n_rows * W
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "threads": cores, "host_cores": host_cores, "kind": "port",
            "sample": f"{n} rays (first {n_rows} rows of the 480x640 frame) in ONE chunk, fwd+bwd to pose, {Nc}+{Ni} samples, "
                      f"8x{Wd} MLP, C={C}, torch {torch.__version__} CPU, {dt:.1f} s; SURVEY 8d names a 32 768-ray chunk "
                      f"(~{32768 / (n / dt):.0f} s at this rate): bounded to the bench contract's 10-30 s of CPU work, rays/s is "
                      f"per-ray work and does not depend on the chunk length; `cores` = `threads` = the torch thread count used "
                      f"(calibrated on a quarter-size run: every hardware thread of a many-core host is slower on 256-wide "
                      f"GEMMs), `host_cores` = os.cpu_count()"}


def committed_traffic(backward, h3, x6):
    """HBM-side bytes per launch of the dominant kernel, from the newest committed counter digest (profiles/rNN/pmc_per_launch.json:
    separate rocprofv3 --pmc passes of this same command, tools/profile_round.sh).  A digest is only quoted while the kernel
    sources it was measured on are the ones in this tree (`_meta.kernel_sources`, tools/pmc_aggregate.py); otherwise the line
    carries traffic = null and says why.  Returns (bytes or None, source string)."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_aggregate import kernel_source_digest
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_per_launch.json")))
    if not files:
        return None, "no committed counter digest under profiles/"
    path = files[-1]
    rel = os.path.relpath(path, ROOT)
    try:
        pm = json.load(open(path))
        have, want_src = pm.get("_meta", {}).get("kernel_sources"), kernel_source_digest(ROOT)
        if have != want_src:
            return None, f"{rel} is STALE: collected on kernel sources {have}, this tree is {want_src} (re-run tools/profile_round.sh)"
        if backward:
            want = "field_bwd_h3_kernel<256,2,0" if h3 else ("field_bwd_kernel<256,19,0,6," if x6 else "field_bwd_kernel<256,19,0,0,")
        else:
            want = "field_fwd_h3_kernel<2,0,256" if h3 else ("field_fwd_x6_kernel<2," if x6 else "field_fwd_kernel<256,1,2")
        e = next(v for k, v in pm.items() if k.replace(" ", "").startswith(want))
        return (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0, f"{rel} (kernel sources {have})"
    except Exception as ex:
        return None, f"{rel}: {ex}"


def _structure_scene(named, gain, decay, sigma_gain):
    """The synthetic scene of tests/golden/refine50*.npz (tools/make_golden_refine50.py applies the same deterministic edit to the
    reference's seed-0 modules): hidden weights x gain, the positional columns of frequency band k of the two layers that read the
    xyz embedding x 2^(-decay k), density head x sigma_gain -- a field with spatial structure instead of a nearly constant one."""
    with torch.no_grad():
        for name, p in named.items():
            if name.startswith(("xyz_encoding_", "dir_encoding", "transient_encoding")) and name.endswith("weight"):
                p.mul_(gain)
        for lname in ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight"):
            w = named[lname]
            for k in range(10):
                w[:, 3 + 6 * k: 9 + 6 * k] *= float(2.0 ** (-decay * k))
        named["static_sigma.0.weight"].mul_(sigma_gain)


def _pose_error(gt, pred):
    """(|t - t'| in metres, rotation angle of R' R^T in degrees): the reference's metric (dm/pose_model.py:75-92 == eval.py:34-51)."""
    import numpy as np
    a, b = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
    R = b[:3, :3] @ a[:3, :3].T
    s_ = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return [float(np.linalg.norm(a[:3, 3] - b[:3, 3])), float(np.degrees(np.arctan2(s_, 0.5 * (np.trace(R) - 1.0))))]


class _TinyAPR(torch.nn.Module):
    """The fixture's stand-in for the absolute-pose-regression CNN (out of scope): Linear(12, 12) on the 2x2 average-pooled image."""

    def __init__(self, weight, bias):
        super().__init__()
        self.fc = torch.nn.Linear(12, 12)
        with torch.no_grad():
            self.fc.weight.copy_(torch.from_numpy(weight))
            self.fc.bias.copy_(torch.from_numpy(bias))

    def forward(self, x):
        B, Cc, H, W = x.shape
        if H % 2 == 0 and W % 2 == 0:       # = adaptive_avg_pool2d(x, 2), whose ROCm kernel takes 1.5 ms for this image (a stand-in
            pooled = x.reshape(B, Cc, 2, H // 2, 2, W // 2).mean((3, 5))          # should not be the largest item of the iteration it times)
        else:
            pooled = torch.nn.functional.adaptive_avg_pool2d(x, 2)
        return self.fc(pooled.reshape(B, -1))


def refinement_loop(dev, iters=50, graph=True, images=1, mode="upsampled", streams=1):
    """BASELINE configs[4] without the DFNet CNN (out of scope) on the scene of tests/golden/refine50_60x80.npz --
    the 60 x 80 rays the reference's own loop renders (DFM_APR_refine.py:107), one perturbed start, the reference's 50-iteration
    results beside it.  Per query image `iters` iterations of pose -> render(80x60) -> affine colour transform -> FusionNet ->
    feature loss -> backward -> Adam, driven by nefes_amd.refine.PoseRefiner:
      mode "3"          DFM_optimization_NFF (DFM_pose_refine.py:290-348): LearnPose, loss at 1/4 resolution -- the fixture's mode 3
      mode "2"          train_on_batch (DFM_APR_refine.py:84-156, the shipped default): the pose is a small regression network's output,
                        loss on the bicubically up-sampled, cropped features -- the fixture's mode 2
      mode "upsampled"  LearnPose with mode 2's up-sampled loss: the timing line of rounds 1-3 (one captured HIP graph per iteration)
    Returns (seconds per image, rays rendered per image, pose errors or None)."""
    import numpy as np
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.refine import PoseRefiner
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "refine50_60x80.npz")))
    T = lambda a_: torch.from_numpy(np.asarray(a_))
    Wd, C = int(g["Wd"]), int(g["C"])
    H, W, focal = (float(v) for v in g["hwf"])
    H, W, ts = int(H), int(W), int(g["tinyscale"])
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True)
    gain, decay, sg = (float(v) for v in g["scene"])
    with torch.no_grad():
        coarse.exposure_embedding.params.copy_(T(g["exposure_params"]))
    for n_ in (coarse, fine):
        _structure_scene(dict(n_.named_parameters()), gain, decay, sg)
    coarse, fine = coarse.requires_grad_(False).to(dev), fine.requires_grad_(False).to(dev)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21,
                                 encode_hist=True)
    kw = dict(network_query_fn=None, perturb=False, N_importance=int(g["Ni"]), N_samples=int(g["Nc"]), network_fn=coarse,
              network_fine=fine, use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False,
              lindisp=False)
    world = dict(pose_scale=float(g["pose_scale"]), pose_scale2=float(g["pose_scale2"]), move_all_cam_vec=g["move_all_cam_vec"].tolist())
    init, hist = T(g["init_c2w"][0]).to(dev), T(g["hist"]).to(dev)
    low = T(g["target_low"])
    full = torch.nn.functional.interpolate(low[None], size=(H, W), mode="bicubic")[0]            # what mode 2 matches against
    common = dict(tinyscale=ts, lr_r=float(g["lr"][0]), lr_t=float(g["lr"][1]), world_setup=world, device=dev)
    if streams > 1:            # refiners on several streams share the frozen networks: no BatchNorm bookkeeping (refine_concurrently)
        common["bn_running_stats"] = False
    n_img = 3
    if mode == "2":
        # the query image, its features and the histogram are resident before anything is timed (as mode 3's target is): the 39 MB
        # feature image copied from the host inside refine_apr cost 0.8 ms per image (tools/time_mode2_parts.py)
        photo, full = (T(g["photo_u8"]).float()[None] / 255.).to(dev), full.to(dev)
        ref = PoseRefiner(kw, args, (H, W, focal), float(g["near"]), float(g["far"]), graph=graph, pose_model=_TinyAPR(g["m2_weight"][0], g["m2_bias"][0]),
                          svd_reg=True, learning_rate=float(g["m2_lr"]), **common)
        pose, _, _ = ref.refine_apr(photo, full, hist, iters)
        if streams > 1:      # `streams` images of the shipped default mode at the same time (refine_apr_concurrently)
            from nefes_amd.refine import refine_apr_concurrently
            refs = [ref] + [PoseRefiner(kw, args, (H, W, focal), float(g["near"]), float(g["far"]), graph=graph,
                                        pose_model=_TinyAPR(g["m2_weight"][0], g["m2_bias"][0]), svd_reg=True, learning_rate=float(g["m2_lr"]), **common)
                            for _ in range(streams - 1)]
            jobs = [(photo, full, hist)] * streams
            refine_apr_concurrently(refs, jobs, iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_img):
            if streams > 1:
                pose = refine_apr_concurrently(refs, jobs, iters)[-1][0]
            else:
                pose, _, _ = ref.refine_apr(photo, full, hist, iters)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n_img / streams
        err = {"hip": _pose_error(g["true_c2w"], pose.cpu().numpy()), "reference": [float(v) for v in g["m2_err"][0]]}
    else:
        up = mode == "upsampled"
        target = (full[:, 10:-10, 10:-10] if up else low).to(dev)
        if images > 1:    # `images` query images refined side by side (PoseRefiner(images=B)); time is reported per image
            init, hist, target = init[None].repeat(images, 1, 1), hist.repeat(images, 1), target[None].repeat(images, 1, 1, 1)
        ref = PoseRefiner(kw, args, (H, W, focal), float(g["near"]), float(g["far"]), upsample=up, graph=graph, images=images, **common)
        pose, _ = ref.refine(init, target, hist, iters)               # packs weights, warms MIOpen, captures the graph
        if streams > 1:     # `streams` query images at the same time, one PoseRefiner + one HIP stream each (refine_concurrently)
            from nefes_amd.refine import refine_concurrently
            refs = [ref] + [PoseRefiner(kw, args, (H, W, focal), float(g["near"]), float(g["far"]), upsample=up, graph=graph, **common)
                            for _ in range(streams - 1)]
            jobs = [(init, target, hist)] * streams
            refine_concurrently(refs, jobs, iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_img):
            if streams > 1:
                pose = refine_concurrently(refs, jobs, iters)[-1][0]
            else:
                pose, _ = ref.refine(init, target, hist, iters)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n_img / images / streams
        p0 = (pose if images == 1 else pose[0])[:3, :4].cpu().numpy()
        err = {"hip": _pose_error(g["true_c2w"], p0)}
        if mode == "3":
            err["reference"] = [float(v) for v in g["m3_err"][0]]
    err["initial"] = [float(v) for v in g["init_err"][0]]
    return sec, iters * (H // ts) * (W // ts), err


def train_steps(dev, steps=10, warmup=2):
    """BASELINE configs[0] on the HIP path: stage-1 colour-only training step of script/run_nefes.py:42-108 --
    200x200 crop (40 000 rays), 64 coarse samples, N_importance=0, reference-default net (8x128, C=128), stratified
    jitter, img2mse loss, backward to the NeRF weights, Adam step.  Returns (seconds per step, rays per step, kernel ms)."""
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render
    H = W = 200
    focal, Wd, C = 525.505 * 200 / 480, 128, 128
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).to(dev)
    prm = [p for n, p in coarse.named_parameters() if not n.startswith(("fusion_net", "exposure_embedding"))]
    opt = torch.optim.Adam(prm, lr=5e-4, betas=(0.9, 0.999))
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1024)
    kw = dict(network_query_fn=None, perturb=1., N_importance=0, N_samples=64, network_fn=coarse, network_fine=None,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    ro, rd, _ = ops.raygen_fwd(H, W, focal, bench_pose().to(dev))                    # get_rays on the GPU
    target = torch.rand(H * W, 3, device=dev)
    losses = []

    def step():
        rgb, _, _, _ = render(H, W, focal, rays=(ro, rd), near=0., far=4., **kw)
        loss = ((rgb - target) ** 2).mean()                              # img2mse
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ops.TIMERS = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    timers, ops.TIMERS = ops.TIMERS, None
    kern = {k: round(sum(s.elapsed_time(e) for s, e in v) / len(v), 4) for k, v in timers.items()}
    # CPU oracle on a bounded slice of the same step (reference settings: netchunk=1024 slices, autograd to the weights)
    from oracle import ref_cpu as O
    th = min(32, os.cpu_count() or 1)
    torch.set_num_threads(th)
    pc = {k: v.requires_grad_() for k, v in O.make_field_params("coarse", Wd, C).items()}
    cfg = O.RenderCfg(N_samples=64, N_importance=0, perturb=1., test_time=False, netchunk=1024)
    n_cpu = 20 * W
    copt = torch.optim.Adam(list(pc.values()), lr=5e-4)
    roc, rdc, tc = ro[:n_cpu].cpu(), rd[:n_cpu].cpu(), target[:n_cpu].cpu()
    t0 = time.perf_counter()
    rgb, _, _, _ = O.render(H, W, focal, pc, None, cfg, rays=(roc, rdc), near=0., far=4.)
    copt.zero_grad()
    ((rgb - tc) ** 2).mean().backward()
    copt.step()
    dtc = time.perf_counter() - t0
    cpu = {"value": n_cpu / dtc, "unit": "rays/s", "cores": th, "kind": "port",
           "sample": f"{n_cpu} rays of the same step (netchunk=1024), torch {torch.__version__} CPU, {dtc:.1f} s"}
    return dt, H * W, kern, [float(l) for l in (losses[0], losses[-1])], cpu


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child and return its exit code."""
    import socket
    import subprocess
    one_gpu = os.environ.get("NEFES_BENCH_ONE_GPU", "0") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_gpu:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible; refusing to run fewer ranks than asked "
              f"(NEFES_BENCH_ONE_GPU=1 NEFES_BENCH_BACKEND=gloo runs all ranks on cuda:0 as a functional check)",
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# What each rank of an N-GPU run executes, measured on ONE GPU (tools/shard_times.sh, round 3: the headline step on the first 1/N of the
# frame's rows): the step has no fixed cost worth naming, so this is the per-rank time a perfect N-GPU run would show on that box.
SHARD_MS_ONE_GPU = {1: 433.3, 2: 216.6, 4: 108.1, 8: 54.2}


def multi_gpu_diagnostics(dist, dev, world, rank, ar_times, a):
    """What a one-shot N-GPU run needs to explain itself: time inside the pose-gradient all-reduce per rank, every
    rank's sustained MFMA clock right after the timed region, the collective library's version and the node's link topology."""
    import ctypes as _C
    from nefes_amd import lib as L
    dev_ms = [s.elapsed_time(e) for s, e, _ in (ar_times or [])]
    host_ms = [h * 1e3 for _, _, h in (ar_times or [])]
    mine = torch.tensor([sum(dev_ms) / max(len(dev_ms), 1), max(dev_ms, default=0.), sum(host_ms) / max(len(host_ms), 1), 0., 0.],
                        device=dev, dtype=torch.float64)
    try:
        ghz, tf = _C.c_double(), _C.c_double()
        L.check(L.load().nefes_probe_mfma_clock(1, 40, _C.byref(ghz), _C.byref(tf), None), "nefes_probe_mfma_clock")
        mine[3], mine[4] = ghz.value, tf.value
    except Exception:
        pass
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    if rank != 0:
        return None
    rows = [[float(v) for v in t.tolist()] for t in every]
    out = {"all_reduce_ms_device_mean_by_rank": [round(r[0], 4) for r in rows], "all_reduce_ms_device_max_by_rank": [round(r[1], 4) for r in rows],
           "all_reduce_ms_host_call_mean_by_rank": [round(r[2], 4) for r in rows], "all_reduces_per_rank": len(dev_ms),
           "sustained_clock_ghz_by_rank": [round(r[3], 3) for r in rows], "sustained_16bit_mfma_tflops_by_rank": [round(r[4], 1) for r in rows],
           "note": "all_reduce device time = events on the compute stream around the call: it includes waiting for the slowest rank to arrive"}
    try:
        out["collective_library"] = {"backend": dist.get_backend(), "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                                     "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                                     "loaded_from": sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "rccl" in ln or "nccl" in ln})[:4]}
    except Exception as e:
        out["collective_library"] = {"error": str(e)}
    try:                                                            # link types / hops between the node's GPUs (a child process; never an exec)
        import subprocess
        topo = []
        for cmd in (["/opt/rocm/bin/rocm-smi", "--showtopo"], ["/opt/rocm/bin/amd-smi", "topology"]):      # (the --csv form prints nothing on this image)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
            topo = [ln.rstrip() for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("WARNING")]
            if len(topo) > 3:
                break
        out["topology"] = topo[:80]
    except Exception as e:
        out["topology"] = f"unavailable: {e}"
    if (a.workload, a.height, a.width) == ("metric", 0, 0) and world in SHARD_MS_ONE_GPU:
        out["one_gpu_shard_ms_reference"] = SHARD_MS_ONE_GPU[world]
        out["one_gpu_shard_ms_reference_source"] = "tools/shard_times.sh, round 3, one box (boxes differ by +-3 %)"
    return out


def hbm_kernel_lines(kern, rays, Nc, Ni, Wd, C):
    """the HBM-bound kernels of the step as GB/s of ALGORITHMIC bytes against the 8 TB/s spec (DESIGN.md 4.2 / 4.3 state the
    byte counts: the forward reads a ray's raw block + depths and writes weights + maps; the backward reads nine rows + depths + the
    upstream maps and writes the gradient block; the coarse pass reads sigma and writes the merged depths)."""
    S = Nc + Ni
    fh = any("fh]" in k for k in kern)
    R = 3 + (Wd // 2 + 1 if fh else C) + 6
    rows_written_bwd = (3 + 1 + 6) if fh else R                 # factored head: the static weight in place of the g rows' products
    maps = (3 + (Wd // 2 + 1 if fh else C) + 3) * 4
    per_ray = {"composite_fwd": S * R * 4 + S * 4 + S * 4 + maps,
               "composite_bwd": 9 * S * 4 + S * 4 + rows_written_bwd * S * 4 + maps,
               "coarse_sample": Nc * 4 + S * 4}
    out = {}
    for k, b in per_ray.items():
        if k in kern and kern[k] > 0:
            gbs = b * rays / (kern[k] * 1e-3) / 1e9
            out[k] = {"ms": round(kern[k], 4), "algorithmic_bytes_per_ray": b, "GB/s": round(gbs, 3), "frac_of_8TB/s": round(gbs / 8000.0, 6)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=0, help="override the workload's frame height (debugging)")
    ap.add_argument("--width", type=int, default=0, help="override the workload's frame width (debugging)")
    ap.add_argument("--cpu-rows", type=int, default=12, help="rows of the frame timed on the host cores (0 = skip)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["loop50", "train"], default="metric")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # Not started by a launcher: start the N ranks ourselves (one process per GPU) and relay rank 0's JSON line.
        # This parent makes no GPU call (torch.cuda.device_count() does not initialise the runtime on this image).
        raise SystemExit(self_launch(a.gpus))
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NEFES_BENCH_ONE_GPU=1 + NEFES_BENCH_BACKEND=gloo: functional test of the N>1 path on a single-GPU box (every rank on
    # cuda:0, collectives staged through the host); never a measurement.
    one_gpu = os.environ.get("NEFES_BENCH_ONE_GPU", "0") == "1"
    dev_index = 0 if (world == 1 or one_gpu) else local_rank
    torch.cuda.set_device(dev_index)
    # NEFES_BENCH_FORCE_GROUP=1: a ONE-rank run still builds the process group and goes through every collective of the N-rank code
    # path (barrier, the pose-gradient all-reduce with its events, the gathers of the diagnostics) -- the RCCL calls of an 8-GPU run
    # exercised on a single-GPU box (tests/test_gpu_a_bench_launch.py); a functional check, never a measurement.
    grouped = world > 1 or os.environ.get("NEFES_BENCH_FORCE_GROUP", "0") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("NEFES_BENCH_BACKEND", "nccl")          # "nccl" = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)

    from nefes_amd import dist as D
    from nefes_amd import lib as L, ops
    D.ONE_RANK_COLLECTIVES = grouped and world == 1
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render

    if a.workload == "train":
        sec, rays, kern, ls, cpu = train_steps(dev)
        # weight-gradient training step: forward + dX + dW = 3 x the forward MACs of the static coarse net
        flop = 3 * 2 * (130944 + 128 * 128 + (128 + 27) * 64 + 64 * 131) * rays * 64
        print(json.dumps({"metric": "rays/s (training step), secondary workload 'train'", "value": rays / sec, "unit": "rays/s",
                          "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
                          "ms_per_step": sec * 1e3, "kernels_ms": kern, "loss_first_last": ls,
                          "algorithmic_tflops": flop / sec / 1e12, "cpu_baseline": cpu,
                          "config": {"workload": "BASELINE configs[0] on the HIP path: stage-1 colour-only training step, 200x200 "
                                                 "crop, 64 coarse samples, N_importance=0, 8x128 net with 128-ch feature head, "
                                                 "perturb=1, img2mse, backward to weights, Adam"}}), flush=True)
        return
    if a.workload == "loop50":
        sec_e, rays, _ = refinement_loop(dev, graph=False)
        sec, rays, err_up = refinement_loop(dev, graph=True)
        sec_b, _, _ = refinement_loop(dev, graph=True, images=8)
        sec_s2, _, err_s2 = refinement_loop(dev, graph=True, streams=2)
        sec_s3, _, _ = refinement_loop(dev, graph=True, streams=3)
        sec3, _, err3 = refinement_loop(dev, graph=True, mode="3")
        sec2e, _, _ = refinement_loop(dev, graph=False, mode="2")
        sec2, _, err2 = refinement_loop(dev, graph=True, mode="2")
        sec2s, _, err2s = refinement_loop(dev, graph=True, mode="2", streams=2)
        sec2s4, _, err2s4 = refinement_loop(dev, graph=True, mode="2", streams=4)
        sec_s4, _, _ = refinement_loop(dev, graph=True, streams=4)
        print(json.dumps({"metric": "rays/s (fwd+bwd), secondary workload 'loop50'", "value": rays / sec, "unit": "rays/s",
                          "n_gpus": 1, "higher_is_better": True,
                          "dtype": "f32",
                          "data": "synthetic", "vs_baseline": None,
                          "ms_per_image_50_iterations": sec * 1e3, "ms_per_image_50_iterations_eager": sec_e * 1e3,
                          "ms_per_image_50_iterations_8_images_side_by_side": sec_b * 1e3,
                          "ms_per_image_50_iterations_2_images_on_2_streams": sec_s2 * 1e3,
                          "ms_per_image_50_iterations_3_images_on_3_streams": sec_s3 * 1e3,
                          "ms_per_image_50_iterations_mode3": sec3 * 1e3, "ms_per_image_50_iterations_mode2": sec2 * 1e3,
                          "ms_per_image_50_iterations_mode2_eager": sec2e * 1e3,
                          "ms_per_image_50_iterations_mode2_2_images_on_2_streams": sec2s * 1e3,
                          "ms_per_image_50_iterations_mode2_4_images_on_4_streams": sec2s4 * 1e3,
                          "ms_per_image_50_iterations_4_images_on_4_streams": sec_s4 * 1e3,
                          "pose_error_m_deg_after_50_iterations": {"mode3": err3, "mode2": err2, "mode2_2_streams": err2s, "mode2_4_streams": err2s4, "upsampled_loss_learnpose": err_up,
                                                                   "reference_from": "tests/golden/refine50_60x80.npz: the reference's own DFM_optimization_NFF / "
                                                                                     "train_on_batch on the CPU from the same start"},
                          "config": {"workload": "BASELINE configs[4] minus the DFNet CNN, on the scene of tests/golden/refine50_60x80.npz: 50 x "
                                                 "[pose -> render 80x60 (64+64, 8x128, C=128) -> affine colour -> FusionNet -> bicubic x4 -> cosine "
                                                 "feature loss -> backward -> Adam]"}}), flush=True)
        return
    wl = WORKLOADS[a.workload]
    Wd, C, Nc, Ni = wl["Wd"], wl["C"], wl["Nc"], wl["Ni"]
    H, W = a.height or wl["H"], a.width or wl["W"]
    focal = wl["focal"] * W / wl["W"]
    near, far = wl["near"], wl["far"]
    in_xyz = 32 if wl["hashgrid"] else 63
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C, in_channels_xyz=in_xyz).requires_grad_(False).to(dev)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, in_channels_xyz=in_xyz, encode_appearance=True,
                     encode_transient=True).requires_grad_(False).to(dev)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21)
    kw = dict(network_query_fn=None, perturb=False, N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
    if wl["hashgrid"]:
        # table scaled to O(0.3) so that the MLP sees the position (tiny-cuda-nn's 1e-4 init would feed it ~zeros)
        kw["xyz_encoder"] = ops.HashGrid(25.0, device=dev)
        kw["xyz_encoder"].table.mul_(3e3)
    pose = bench_pose().to(dev)
    row0, nrows = D.row_shard(H, rank, world)
    n_total = H * W

    def step():
        c2w = pose.clone().requires_grad_()
        rgb, disp, acc, ex = render(H, W, focal, c2w=D.replicate_pose(c2w), near=near, far=far, row_range=(row0, nrows), **kw)
        feat = ex["feat_map"]
        loss = (feat ** 2).sum() / (n_total * C) + (rgb ** 2).sum() / (n_total * 3)   # = mean over the full frame
        loss.backward()                                                               # pose all-reduce happens here
        return c2w.grad

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    ops.TIMERS = {}
    D.ALLREDUCE_TIMES = [] if grouped else None
    t0 = time.perf_counter()
    trace = os.environ.get("NEFES_BENCH_TRACE", "0") == "1"           # debugging: host time of every step (adds a sync per step)
    for _ in range(a.steps):
        ts = time.perf_counter()
        g = step()
        if trace:
            if os.environ.get("NEFES_BENCH_TRACE_SYNC", "1") == "1":
                torch.cuda.synchronize()
            print(f"step {time.perf_counter() - ts:.4f} s", file=sys.stderr, flush=True)
    t_enq = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if trace:
        print(f"enqueue {t_enq:.4f} s, with the final sync {dt:.4f} s", file=sys.stderr, flush=True)
    timers, ops.TIMERS = ops.TIMERS, None
    ar_times, D.ALLREDUCE_TIMES = D.ALLREDUCE_TIMES, None
    rank_ms = [dt / a.steps * 1e3]
    multi = None
    if grouped:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)                                  # per-rank step time (min / max over ranks in the line)
        rank_ms = [float(t.item()) / a.steps * 1e3 for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        multi = multi_gpu_diagnostics(dist, dev, world, rank, ar_times, a)

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = n_total * a.steps / dt
        kern = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in timers.items()}   # ms per launch, rank 0
        rays_local = nrows * W
        # Dominant kernel = the longest-running launch of the step.  Fine-field forward (FULL mode) and backward-to-inputs
        # do the same algorithmic MACs per sample.  roofline.achieved = algorithmic fp32 FLOPs per launch / launch time.
        # Peak: a plain fp32-MFMA kernel is bounded by 157.3 TFLOP/s; a bf16x6 kernel executes six bf16 MFMA FLOPs per
        # algorithmic FLOP, so its bound is the dense bf16 MFMA peak / 6 = 416.7 TFLOP/s of algorithmic work.
        flop_fwd = 2.0 * macs_full(Wd, C, in_xyz) * rays_local * (Nc + Ni)
        fwd_key = next(k for k in kern if k.startswith("field_fwd[full"))
        bwd_key = next(k for k in kern if k.startswith("field_bwd"))
        dom_key = max((fwd_key, bwd_key), key=lambda k: kern[k])
        h3 = ",h3" in dom_key or "[h3" in dom_key         # fp16 two-part split products: three MFMAs per algorithmic product
        grid_in_kernel = "hashgrid" in dom_key            # configs[3], round 5: the field kernels gather the hash grid themselves
        x6 = dom_key.endswith("x6]")
        ach = flop_fwd / (kern[dom_key] * 1e-3) / 1e12
        enc = 2 if grid_in_kernel else int(wl['hashgrid'])
        if dom_key == bwd_key:
            dom_name = (f"field_bwd_h3_kernel<{Wd},{2 if 3 + C <= 32 else 9},{enc}>" if h3 else
                        f"field_bwd_kernel<{Wd},{3 + C},{enc}{',X6' if x6 else ''}>")
        else:
            dom_name = (f"field_fwd_h3_kernel<FULL,{enc},{Wd},{1 if 3 + C <= 32 else 5}>" if h3 else
                        "field_fwd_x6_kernel<FULL>" if x6
                        else f"field_fwd_kernel<{Wd},{(3 + C + 31) // 32},FULL,{enc}>")
        if "fh]" in dom_key:                              # factored feature head (round 5): the 3-row colour head's instance, FH = true
            dom_name = (f"field_bwd_h3_kernel<{Wd},2,0,FH>" if dom_key == bwd_key else f"field_fwd_h3_kernel<FULL,0,{Wd},1,FH>")
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if h3 else (PEAK_BF16_MFMA_TFLOPS / 6.0 if x6 else PEAK_F32_MFMA_TFLOPS)
        flop_frame = 2.0 * (Nc * macs_sigma(Wd, in_xyz) + 2 * (Nc + Ni) * macs_full(Wd, C, in_xyz)) * n_total
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the figure
        # comes from the committed rocprofv3 passes of this same command (profiles/, 2*FETCH_SIZE + WRITE_SIZE in KiB,
        # gfx950 correction of MI355X_MICROARCH.md) and is quoted only for the workload it was measured on.
        traffic, traffic_source = None, None
        if (a.workload, H, W, world) == ("metric", 480, 640, 1):
            traffic, traffic_source = committed_traffic(dom_key == bwd_key, h3, x6)
        out = {
            "metric": "rays/s (fwd+bwd) at 640x480x(64+128) samples, 8x256 MLP" if a.workload == "metric"
                      else f"rays/s (fwd+bwd), secondary workload '{a.workload}'", "value": value, "unit": "rays/s",
            "n_gpus": world, "world_size": dist.get_world_size() if grouped else 1,
            "collective_backend": (dist.get_backend() if grouped else None), "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "matrix_core_arithmetic": "fp16 two-part split (3 products)" if h3 else ("bf16x6" if x6 else "fp32 MFMA"),
            "data": "synthetic",
            "config": {"workload": f"{wl['name']}; {W}x{H}, random seed-0 weights, fwd + bwd to the 3x4 pose",
                       "rays_per_step": n_total, "samples_per_ray": [Nc, Ni], "parallelism": f"rows/{world}",
                       "arithmetic": ("fp32 in, fp32 out; matrix products as fp16 two-part split products (hi, lo fp16 pairs of power-of-two "
                                      "scaled operands, 22 significant bits, three cross terms, fp32 accumulation: fp32-level accuracy, "
                                      "tests/test_gpu_h3.py); NEFES_SPLIT=x6 / f32 select the bf16x6 / fp32-MFMA kernels"
                                      if h3 else
                                      "fp32 in, fp32 out; matrix products as exact bf16x6 split products with fp32 accumulation "
                                      "(fp32-level accuracy, tests/test_gpu_x6.py); NEFES_X6=0 selects the fp32-MFMA kernels"
                                      if any(k.endswith("x6]") for k in kern) else "fp32 MFMA")},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": ach,
                         "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "peak_basis": ("dense fp16 MFMA peak 2500 / 3: fp16 two-part split, three products per algorithmic product, "
                                        "fp32-level accuracy" if h3 else
                                        "dense bf16 MFMA peak 2500 / 6: bf16x6 split products, fp32-level accuracy" if x6
                                        else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                         "vs_fp32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_unit": "bytes/launch (HBM side: 2 x FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 PMC passes, profiles/)",
                         # whole step against the same bound as the dominant kernel (weak: the step is 99.9 % these kernels)
                         "end_to_end_frac": value * (flop_frame / n_total) / (peak * 1e12 * world),
                         "end_to_end_vs_fp32_mfma_peak": value * (flop_frame / n_total) / (PEAK_F32_MFMA_TFLOPS * 1e12 * world),
                         "hbm_kernels": hbm_kernel_lines(kern, rays_local, Nc, Ni, Wd, C)},
            "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items())},
            "pose_grad_abs_max": float(g.abs().max()),
            # the all-reduced 3x4 pose gradient itself (row-major), so that an N-rank run can be compared with the 1-rank run number
            # by number (tests/test_gpu_a_bench_launch.py), and each rank's own step time
            "pose_grad": [float(v) for v in g.detach().reshape(-1).cpu().tolist()],
            "ms_per_step_by_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": [round(t, 4) for t in rank_ms]},
        }
        if multi is not None:
            out["multi_gpu"] = multi
            if "one_gpu_shard_ms_reference" in multi:
                out["multi_gpu"]["scaling_efficiency_vs_one_gpu_shard"] = multi["one_gpu_shard_ms_reference"] / ms_step
        if h3 or x6:
            # The 2 500 TFLOP/s data-sheet peak is a 2.4 GHz figure; under sustained MFMA issue with operands whose bits toggle
            # the chip's power management holds a lower clock.  Measured here, on this box, right after the timed region
            # (csrc/probe.hip: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD for ~40 ms; fp16 and bf16 MFMAs issue alike).
            try:
                import ctypes as _C
                ghz, tf = _C.c_double(), _C.c_double()
                L.check(L.load().nefes_probe_mfma_clock(1, 40, _C.byref(ghz), _C.byref(tf), None), "nefes_probe_mfma_clock")
                per = 3.0 if h3 else 6.0
                out["roofline"]["sustained"] = {
                    "clock_ghz": round(ghz.value, 3), "dense_16bit_mfma_tflops": round(tf.value, 1),
                    "peak": tf.value / per, "frac": ach / (tf.value / per),
                    "basis": "back-to-back v_mfma_f32_32x32x16_f16 with random operands on all CUs, measured in this run "
                             "(power-limited clock; the data-sheet peak assumes 2.4 GHz)"}
            except Exception as e:                       # the probe is an aid, never a reason to lose the bench line
                out["roofline"]["sustained"] = {"error": str(e)}
        if world == 1 and a.cpu_rows > 0 and a.workload in ("metric", "metric128"):
            out["cpu_baseline"] = cpu_baseline(Wd, C, Nc, Ni, a.cpu_rows, W, focal)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
