#!/usr/bin/env python3
"""Generate tests/golden/refine50.npz: BASELINE configs[4] -- the 50-iteration analysis-by-synthesis loop -- executed by the
REFERENCE'S OWN LOOP FUNCTIONS on the CPU of this container (the reference never travels; these vectors do), on a synthetic
population of K = 8 perturbed initial poses around one ground-truth camera, with the pose-error metric of script/eval.py.

Both refinement modes, each CALLED AS IT IS in /root/reference (round 2's fixture re-stated the call sequence instead):

    mode 3  (`--pose_only 3`)                 dm.DFM_pose_refine.DFM_optimization_NFF     script/dm/DFM_pose_refine.py:290-348
        LearnPose (se(3) delta, lietorch=False) -> fix_coord_supp -> render -> affine_color_transform -> run_fusion_net
        -> feature_loss at 1/tinyscale resolution -> backward -> Adam(lr_r, lr_t)                     (args per dm/options.py:137-139)
    mode 2  (`pose_only=2`, the shipped default, config_stairs_DFM.txt:21)   dm.DFM_APR_refine.train_on_batch   script/dm/DFM_APR_refine.py:84-156
        inference_pose_regression(model) + svd_reg -> fix_coord_supp -> render -> affine_color_transform -> run_fusion_net
        -> nn.Upsample(size=(H, W), mode='bicubic') -> 10 px crop -> FeatureLoss -> backward -> Adam over the regression
        network's parameters; PSNR / SSIM of the verification step; after the loop the outer function's inference + roll-back
        rule (:236-250), transcribed in `after_loop` below because `DFM_post_processing` itself loads checkpoints and a dataset

The only things changed around the reference's code, and why:
  * `torch.set_default_device` is replaced by a no-op for the duration of the run: both loop functions switch torch's default
    device to 'cuda' (DFM_pose_refine.py:316, DFM_APR_refine.py:103) and this container has no GPU;
  * third-party modules the container lacks are import stubs (tools/make_golden_refine.py); two of them carry arithmetic:
    `tinycudann.Network` (the exposure network: a plain fp32 MLP with the flat parameter layout of nefes_amd.field.ExposureMLP --
    tiny-cuda-nn's own layout and fp16 arithmetic stay UNPINNED) and `cv2.Rodrigues` (scipy's rotation vector), which
    `dm.pose_model.compute_pose_error_SE3` (== eval.py:34-51) needs for the rotation error;
  * the regression network of mode 2 (DFNet: a CNN outside the path) is `TinyAPR`: Linear(12, 12) on the 2x2 average-pooled query
    image, `PoseEstimatorType='PoseNet'` so that `inference_pose_regression` calls it as `model(inputs)` and applies svd_reg.  What
    the path needs from it is that the pose is a function of trainable parameters;
  * the loss value and the gradients are captured from the outside (a wrapper around `feature_loss`, an optimizer pre-step hook, a
    tensor hook on the network output); the reference's functions do not return them.

The scene (oracle.refine_cpu.structure_scene): seed-0 random weights edited deterministically so that the field has spatial
structure with a falling spectrum -- round 2's default-init scene was nearly constant in space, its translation gradient was
rounding noise that Adam amplified, and the free-running comparison was ill-conditioned.  Here every start converges: the K
refined poses end at ~1e-2 m / ~0.3 deg from a 0.08-0.16 m / 2-5 deg initial error, which is the regime of the reference's published
7-Scenes numbers (DFNet 0.12 m / 2.9 deg -> DFNet + NeFeS50 0.05 m / 1.3 deg on stairs).

Also stored, clearly labelled as ORACLE output (not the reference's): the float64 run of oracle/refine_cpu.py from the same
starts (`*_f64`), which the tests use as the truth of the three-way rule (e_ref = reference fp32 vs float64).

Usage:  python tools/make_golden_refine50.py [--threads 8] [--skip-f64]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import make_golden_refine as G                                                    # noqa: E402  (stubs, FlatMLP, rot)
from oracle import ref_cpu as O                                                   # noqa: E402
from oracle import refine_cpu as RC                                               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SCENE = dict(gain=3.0, decay=1.0, sigma_gain=4.0)
K, ITERS = 8, 50
STRETCH = torch.tensor([[1.06, 0.02, -0.01], [0.02, 0.95, 0.015], [-0.01, 0.015, 1.01]])


class TinyAPR(torch.nn.Module):
    """Stand-in for the absolute-pose-regression CNN: [1,3,H,W] -> [1,12].  `hook` receives d loss / d output."""

    def __init__(self, weight, bias, hook=None):
        super().__init__()
        self.fc = torch.nn.Linear(12, 12)
        with torch.no_grad():
            self.fc.weight.copy_(weight)
            self.fc.bias.copy_(bias)
        self.hook = hook

    def forward(self, x):
        out = self.fc(torch.nn.functional.adaptive_avg_pool2d(x, 2).reshape(x.shape[0], -1))
        if self.hook is not None and out.requires_grad:
            out.register_hook(self.hook)
        return out


def npy(t):
    return t.detach().cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--skip-f64", action="store_true")
    ap.add_argument("--k", type=int, default=K, help="dry runs only: fewer starts")
    ap.add_argument("--iters", type=int, default=ITERS, help="dry runs only: fewer iterations")
    ap.add_argument("--out", default=os.path.join(OUT, "refine50.npz"))
    ap.add_argument("--hw", type=float, nargs=3, default=[120, 160, 150.0], metavar=("H", "W", "FOCAL"),
                    help="full-resolution image; the loop renders (H/4) x (W/4) rays.  240 320 300 = the 60 x 80 rays of the reference's own "
                         "loop (DFM_APR_refine.py:107, seven_scenes_colmap.py:264-276): `--hw 240 320 300 --k 1 --out tests/golden/refine50_60x80.npz`")
    a = ap.parse_args()
    globals().update(K=a.k, ITERS=a.iters)
    torch.set_num_threads(a.threads)
    DR, M = G.import_reference()
    import cv2                                                                    # the import stub
    from scipy.spatial.transform import Rotation
    cv2.Rodrigues = lambda R: (Rotation.from_matrix(np.asarray(R, dtype=np.float64)).as_rotvec().reshape(3, 1), None)
    import dm.DFM_APR_refine as AR
    import dm.pose_model as PM
    PM.cv2 = cv2
    torch.set_default_device = lambda *args_, **kw_: None                         # see the module docstring

    Wd, C, Nc, Ni = 128, 128, 64, 64          # the refinement shape of the reference (8x128 MLP, 128 feature channels, 64+64)
    H, W, focal, ts = int(a.hw[0]), int(a.hw[1]), float(a.hw[2]), 4      # 30x40 rays by default
    near, far = 0.0, 4.0                      # data/7Scenes/stairs/world_setup.json:2-3
    coarse = M.NeRFH_NFF('coarse', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27, f_dim=C)
    fine = M.NeRFH_NFF('fine', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True,
                       encode_transient=True, in_channels_a=50, in_channels_t=20, f_dim=C)
    g = torch.Generator().manual_seed(5050)
    expo = (torch.rand(coarse.exposure_embedding.params.numel(), generator=g) - 0.5) * 0.5      # colour kernel entries of order 1
    with torch.no_grad():
        coarse.exposure_embedding.params.copy_(expo)
        for net in (coarse, fine):
            RC.structure_scene(dict(net.named_parameters()), **SCENE)
    coarse.requires_grad_(False)
    fine.requires_grad_(False)

    embed_fn, _, _ = M.get_embedder(10, 0, -1)
    embeddirs_fn, _, _ = M.get_embedder(4, 0, -1)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21,
                                 encode_hist=True, tinyscale=ts, chunk=1 << 15, lr_r=0.0087, lr_t=0.01,   # dm/options.py:137-138 (7-Scenes)
                                 PoseEstimatorType='PoseNet', svd_reg=True, batch_size=1, learning_rate=1e-3, per_pixel=False)
    q = lambda inputs, viewdirs, ts_, network_fn, typ, output_transient, test_time, store_rgb: \
        M.run_network_NeRFH_NFF(inputs, viewdirs, ts_, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, typ=typ,
                                output_transient=output_transient, netchunk=args.netchunk, test_time=test_time,
                                store_rgb=store_rgb)
    kw = dict(network_query_fn=q, perturb=0., N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False,
              near=near, far=far)
    world = {"pose_scale": 0.8, "pose_scale2": 1.25, "move_all_cam_vec": [0.1, -0.05, 0.2]}
    hist = torch.tensor([[3., 7., 12., 20., 31., 18., 9., 4., 2., 1.]])
    hwf = (H, W, focal)
    h, w = H // ts, W // ts

    true_c2w = torch.eye(4)
    true_c2w[:3, :3] = G.rot((0.2, 1.0, -0.1), 8.0)
    true_c2w[:3, 3] = torch.tensor([0.15, -0.10, 0.30])

    # ---- the query image's features and the query image, from the ground-truth pose (the reference takes both from a camera +
    # DFNet): the low-resolution fused features are what is stored; every consumer up-samples them with torch on the CPU ----------
    with torch.no_grad():
        pose = DR.fix_coord_supp(args, true_c2w[None, :3, :4].clone(), world, device=None)
        rgb, _, _, extras = DR.render(h, w, focal / ts, chunk=args.chunk, c2w=pose[0, :3, :4], img_idx=hist, **kw)
        rgb = coarse.affine_color_transform(args, rgb, hist, 1)
        render_rgb, _, target_low = coarse.run_fusion_net(rgb, extras['feat_map'], h, w, 1)
        target_low = target_low.clone()                                           # [1,C,h,w]
        target_full = torch.nn.Upsample(size=(H, W), mode='bicubic')(target_low)  # [1,C,H,W]: what mode 2 matches against
        photo_u8 = (torch.nn.Upsample(size=(H, W), mode='bicubic')(render_rgb).clamp(0, 1) * 255).round().to(torch.uint8)
        photo = photo_u8.float() / 255.                                           # [1,3,H,W], as a dataloader hands it over
    print("target features: std over pixels", float(target_low.std((2, 3)).mean()))

    # ---- the population of starts ------------------------------------------------------------------------------------------
    inits = []
    for k in range(K):
        axis = torch.randn(3, generator=g).tolist()
        ang = 2.0 + 3.0 * float(torch.rand((), generator=g))
        dt = torch.randn(3, generator=g)
        dt = dt / dt.norm() * (0.08 + 0.08 * float(torch.rand((), generator=g)))
        c = torch.eye(4)
        c[:3, :3] = G.rot(axis, ang) @ true_c2w[:3, :3]
        c[:3, 3] = true_c2w[:3, 3] + dt
        inits.append(c)
    inits = torch.stack(inits)
    err = lambda p: PM.compute_pose_error_SE3(true_c2w[:3, :4].clone(), torch.as_tensor(np.asarray(p, dtype=np.float32)))
    out = dict(Wd=Wd, C=C, Nc=Nc, Ni=Ni, hwf=np.array(hwf), tinyscale=ts, near=near, far=far, hist=npy(hist), exposure_params=npy(expo),
               scene=np.array([SCENE["gain"], SCENE["decay"], SCENE["sigma_gain"]]), pose_scale=world["pose_scale"],
               pose_scale2=world["pose_scale2"], move_all_cam_vec=np.array(world["move_all_cam_vec"]), true_c2w=npy(true_c2w),
               init_c2w=npy(inits), target_low=npy(target_low[0]), photo_u8=npy(photo_u8[0]),
               init_err=np.array([err(c[:3, :4]) for c in inits]))

    # ---- mode 3: DFM_optimization_NFF, K starts x 50 iterations ----------------------------------------------------------------
    rec = {}
    ref_loss = DR.feature_loss

    def recording_loss(*aa, **kk):
        v = ref_loss(*aa, **kk)
        rec["loss"].append(float(v))
        return v
    DR.feature_loss = recording_loss
    data = photo.clone()
    m3 = {k_: [] for k_ in ("loss", "grad", "r", "t", "pose", "err")}
    t0 = time.time()
    for k in range(K):
        net = DR.LearnPose(1, True, True, inits[k][None].clone(), lietorch=False)
        opt = torch.optim.Adam([{'params': net.r, 'lr': args.lr_r}, {'params': net.t, 'lr': args.lr_t}])   # DFM_post_processing2 :392-398
        rec.update(loss=[], grad=[])
        opt.register_step_pre_hook(lambda o, a_, k_: rec["grad"].append(np.concatenate([npy(net.r.grad[0]), npy(net.t.grad[0])])))
        rs, tsv = [], []
        for it in range(ITERS):
            net = DR.DFM_optimization_NFF(args, 0, data, net, hist, hwf, opt, 'cpu', world, target_low, kw, None)
            rs.append(npy(net.r[0]).copy())
            tsv.append(npy(net.t[0]).copy())
        with torch.no_grad():
            final = npy(net(cam_id=0)[:3, :4])
        for k_, v in (("loss", rec["loss"]), ("grad", rec["grad"]), ("r", rs), ("t", tsv)):
            m3[k_].append(np.asarray(v, dtype=np.float32))
        m3["pose"].append(final)
        m3["err"].append(err(final))
        print(f"mode 3 start {k}: init err {out['init_err'][k]}  ->  {m3['err'][-1]}   loss {rec['loss'][0]:.5f} -> {rec['loss'][-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    DR.feature_loss = ref_loss
    for k_, v in m3.items():
        out["m3_" + k_] = np.stack(v)

    # ---- mode 2: train_on_batch, K starts x 50 iterations ------------------------------------------------------------------
    desc = RC.image_descriptor(photo)
    wgen = torch.Generator().manual_seed(77)
    m2 = {k_: [] for k_ in ("weight", "bias", "loss", "grad", "psnr", "ssim", "pose", "final", "retreat", "err", "w_traj", "b_traj")}
    gt_pose = true_c2w[:3, :4].reshape(1, 12)
    loss_mod = DR.FeatureLoss(per_pixel=args.per_pixel)
    for k in range(K):
        weight = 0.05 * torch.randn(12, 12, generator=wgen)
        # the network's first prediction: the k-th perturbed pose with a rotation block that is NOT orthonormal, as a regression
        # network's raw output is (R S with S symmetric positive definite: svd_reg recovers R; with an exact rotation the three
        # singular values coincide and torch.svd's backward is 0/0)
        raw = inits[k][:3, :4].clone()
        raw[:, :3] = raw[:, :3] @ STRETCH
        bias = raw.reshape(12) - weight @ desc
        rec.update(loss=[], grad=[])
        base = TinyAPR(weight, bias)
        import copy
        pp = copy.deepcopy(base)                                       # DFM_post_processing :209
        pp.hook = lambda gr: rec["grad"].append(npy(gr[0]).copy())
        opt = torch.optim.Adam(pp.parameters(), lr=args.learning_rate)  # :212
        floss = lambda x, y: (lambda v: (rec["loss"].append(float(v)), v)[1])(loss_mod(x, y))
        ps, ss, poses, wt, bt = [], [], [], [], []
        for it in range(ITERS):
            _, psnr_i, ssim_i = AR.train_on_batch(args, data, pp, None, target_full, gt_pose, hist, hwf, opt, 'cpu', world, kw, floss, it)
            ps.append(float(psnr_i))
            ss.append(float(ssim_i))
            wt.append(npy(pp.fc.weight).copy())
            bt.append(npy(pp.fc.bias).copy())
            with torch.no_grad():
                poses.append(npy(DR.inference_pose_regression(args, data, 'cpu', pp)[0]))
        # after_loop: DFM_post_processing :233-250 (inference of the refined network; roll back when PSNR or SSIM got worse)
        with torch.no_grad():
            predict = DR.inference_pose_regression(args, data, 'cpu', pp).reshape(1, 3, 4)
            retreat = bool(ps[-1] < ps[0]) or bool(ss[-1] < ss[0])
            if retreat:
                predict = DR.inference_pose_regression(args, data, 'cpu', base).reshape(1, 3, 4)
        for k_, v in (("weight", npy(weight)), ("bias", npy(bias)), ("loss", np.asarray(rec["loss"], np.float32)),
                      ("grad", np.stack(rec["grad"])), ("psnr", np.asarray(ps)), ("ssim", np.asarray(ss)), ("pose", np.stack(poses)),
                      ("final", npy(predict[0])), ("retreat", retreat), ("err", err(npy(predict[0]))),
                      ("w_traj", np.stack(wt)), ("b_traj", np.stack(bt))):
            m2[k_].append(v)
        print(f"mode 2 start {k}: init err {out['init_err'][k]}  ->  {m2['err'][-1]}  retreat {retreat}  psnr {ps[0]:.2f} -> {ps[-1]:.2f}  "
              f"loss {rec['loss'][0]:.5f} -> {rec['loss'][-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    for k_, v in m2.items():
        if k_ in ("w_traj", "b_traj"):
            v = v[:2]                                                  # the parameter trajectories of two starts (teacher forcing)
        out["m2_" + k_] = np.stack(v)
    out["m2_lr"] = args.learning_rate
    out["lr"] = np.array([args.lr_r, args.lr_t])
    for tag in ("m3", "m2"):
        e = out[tag + "_err"]
        print(f"{tag}: median error {np.median(e[:, 0]):.5f} m, {np.median(e[:, 1]):.4f} deg   (initial {np.median(out['init_err'][:, 0]):.4f} m, "
              f"{np.median(out['init_err'][:, 1]):.3f} deg)")
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, "(reference part)")

    if a.skip_f64:
        return
    add_f64(a.out, a.iters)


def problem(g, dtype, k, mode):
    """oracle.refine_cpu.Problem of start k (shared with tests/test_refine50_oracle.py)."""
    from nefes_amd.field import NeRFH_NFF
    Wd, C = int(g["Wd"]), int(g["C"])
    net = NeRFH_NFF('coarse', W=Wd, f_dim=C)                       # seed-0 init == the reference's (tests/test_pose.py checksums)
    fsd = {k_: v.detach().clone() for k_, v in net.fusion_net.state_dict().items()}
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    gain, decay, sg = (float(v) for v in g["scene"])
    for p in (pc, pf):
        RC.structure_scene(p, gain, decay, sg)
    cfg = O.RenderCfg()
    cfg.N_samples, cfg.N_importance = int(g["Nc"]), int(g["Ni"])
    world = dict(pose_scale=float(g["pose_scale"]), pose_scale2=float(g["pose_scale2"]), move_all_cam_vec=g["move_all_cam_vec"].tolist())
    H, W, focal = g["hwf"].tolist()
    low = torch.from_numpy(g["target_low"])
    if mode == 2:
        target = torch.nn.functional.interpolate(low[None], size=(int(H), int(W)), mode="bicubic")[0]
    else:
        target = low
    return RC.Problem(pc, pf, fsd, torch.from_numpy(g["exposure_params"]), cfg, (H, W, focal), int(g["tinyscale"]), float(g["near"]),
                      float(g["far"]), torch.from_numpy(g["init_c2w"][k]), target, torch.from_numpy(g["hist"]), world, dtype=dtype,
                      upsample=(int(H), int(W)) if mode == 2 else None)


def add_f64(path, iters=ITERS):
    """The float64 oracle's run from the same starts: `m3_*_f64`, `m2_*_f64` (ORACLE output, the tests' truth)."""
    ITERS = iters
    g = dict(np.load(path))
    K_ = len(g["init_c2w"])
    photo = torch.from_numpy(g["photo_u8"]).float()[None] / 255.
    t0 = time.time()
    p3, r3, t3, p2 = [], [], [], []
    for k in range(K_):
        b = RC.refine(problem(g, torch.float64, k, 3), float(g["lr"][0]), float(g["lr"][1]), ITERS)
        p3.append(b["poses"][-1].numpy())
        r3.append(b["r"].numpy())
        t3.append(b["t"].numpy())
        print(f"f64 mode 3 start {k}: |ref - f64| final pose {np.abs(p3[-1] - g['m3_pose'][k]).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
    g.update(m3_pose_f64=np.stack(p3), m3_r_f64=np.stack(r3), m3_t_f64=np.stack(t3))
    np.savez_compressed(path, **g)
    for k in range(K_):
        b = RC.refine_apr(problem(g, torch.float64, k, 2), torch.from_numpy(g["m2_weight"][k]), torch.from_numpy(g["m2_bias"][k]), photo,
                          float(g["m2_lr"]), ITERS)
        p2.append(b["final"].numpy())
        print(f"f64 mode 2 start {k}: |ref - f64| final pose {np.abs(p2[-1] - g['m2_final'][k]).max():.2e}  retreat {bool(b['retreat'])}  ({time.time() - t0:.0f} s)", flush=True)
    g.update(m2_final_f64=np.stack(p2))
    np.savez_compressed(path, **g)
    print("wrote the float64 oracle runs into", path)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--f64-only":
        torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
        add_f64(os.path.join(OUT, "refine50.npz"))
    else:
        main()
