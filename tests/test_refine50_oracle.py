"""The refinement-loop oracle against tests/golden/refine50.npz: BASELINE configs[4], 50 iterations x 8 perturbed starts, executed
by the REFERENCE's own `DFM_optimization_NFF` (pose_only 3) and `train_on_batch` (pose_only 2, the shipped default) on the CPU
(tools/make_golden_refine50.py), with the reference's pose-error metric (dm/pose_model.py:75-92 == eval.py:34-51).  CPU only; the
GPU twin is tests/test_gpu_refine50.py."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from oracle import refine_cpu as RC


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def problem(g, dtype, k, mode):
    """oracle.refine_cpu.Problem of start k: mode 3 matches at 1/tinyscale resolution, mode 2 at full resolution (bicubic
    up-sampling of the stored low-resolution target by torch on the CPU, exactly what the generator fed the reference)."""
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
    target = torch.nn.functional.interpolate(low[None], size=(int(H), int(W)), mode="bicubic")[0] if mode == 2 else low
    return RC.Problem(pc, pf, fsd, torch.from_numpy(g["exposure_params"]), cfg, (H, W, focal), int(g["tinyscale"]), float(g["near"]),
                      float(g["far"]), torch.from_numpy(g["init_c2w"][k]), target, torch.from_numpy(g["hist"]), world, dtype=dtype,
                      upsample=(int(H), int(W)) if mode == 2 else None)


def photo_of(g):
    return torch.from_numpy(g["photo_u8"]).float()[None] / 255.


def test_pose_error_restatement_matches_the_reference_metric(golden):
    """oracle pose_error == the reference's compute_pose_error_SE3 on every stored pose (the generator called the reference's
    function; cv2.Rodrigues, which this container lacks, was scipy's rotation vector)."""
    g = golden("refine50")
    gt = g["true_c2w"]
    for poses, errs in ((g["init_c2w"], g["init_err"]), (g["m3_pose"], g["m3_err"]), (g["m2_final"], g["m2_err"])):
        for p, e in zip(poses, errs):
            t, r = RC.pose_error(gt, p)
            assert abs(t - e[0]) < 1e-6 * max(e[0], 1e-3) and abs(r - e[1]) < 1e-4 * max(e[1], 1e-2), (t, r, e)


def test_fixture_is_a_converging_population(golden):
    """What makes this fixture a usable stand-in for 'median pose error on 7-Scenes': every start converges (errors fall by > 3x in
    both modes), and the loop is well conditioned -- the float64 oracle run from the same starts ends where the reference's fp32
    run ends, closely enough that the MEDIAN errors of the two populations agree within 1 %."""
    g = golden("refine50")
    init = np.median(g["init_err"], 0)
    for tag, key in (("m3", "m3_pose"), ("m2", "m2_final")):
        med = np.median(g[tag + "_err"], 0)
        assert med[0] < init[0] / 3 and med[1] < init[1] / 3, (tag, med, init)
        e64 = np.array([RC.pose_error(g["true_c2w"], p) for p in g[key + "_f64"]])
        med64 = np.median(e64, 0)
        assert abs(med64[0] - med[0]) < 0.01 * med[0] and abs(med64[1] - med[1]) < 0.01 * med[1], (tag, med, med64)
    assert not g["m2_retreat"].any()            # PSNR and SSIM improve in every run: the verification step keeps the refined pose


@pytest.mark.parametrize("k,its", [(0, (0, 1, 20, 49)), (5, (0, 35))])
def test_oracle_mode3_iteration_matches_reference(golden, k, its):
    """Teacher-forced, `DFM_optimization_NFF`: at the (r, t) the reference held before iteration i, the fp32 oracle returns the
    reference's loss and its gradient to (r, t)."""
    g = golden("refine50")
    p = problem(g, torch.float32, k, 3)
    for i in its:
        r0 = np.zeros(3, np.float32) if i == 0 else g["m3_r"][k, i - 1]
        t0 = np.zeros(3, np.float32) if i == 0 else g["m3_t"][k, i - 1]
        loss, grad = p.loss_and_grad(r0, t0)
        ref_l = float(g["m3_loss"][k, i])
        assert abs(float(loss) - ref_l) < 2e-4 * ref_l + 2e-7, (k, i, float(loss), ref_l)
        assert rel(grad.numpy(), g["m3_grad"][k, i]) < 2e-4, (k, i, grad.numpy(), g["m3_grad"][k, i])


@pytest.mark.parametrize("k,its", [(0, (0, 1, 20, 49)), (1, (0, 35))])
def test_oracle_mode2_iteration_matches_reference(golden, k, its):
    """Teacher-forced, `train_on_batch`: with the regression network's parameters as the reference held them before iteration i,
    the fp32 oracle returns the reference's loss, its gradient to the network's twelve outputs (through svd_reg, fix_coord_supp,
    the render, the colour transform, the fusion CNN, the bicubic up-sampling and the crop), and the verification step's PSNR /
    SSIM."""
    g = golden("refine50")
    p = problem(g, torch.float32, k, 2)
    photo = photo_of(g)
    desc = RC.image_descriptor(photo)
    for i in its:
        W = torch.from_numpy(g["m2_weight"][k] if i == 0 else g["m2_w_traj"][k, i - 1]).clone().requires_grad_()
        b = torch.from_numpy(g["m2_bias"][k] if i == 0 else g["m2_b_traj"][k, i - 1]).clone().requires_grad_()
        raw = W @ desc + b
        loss, img = p.loss_at_pose(RC.svd_reg(raw.reshape(3, 4)), want_rgb=True)
        graw, = torch.autograd.grad(loss, raw)
        ref_l = float(g["m2_loss"][k, i])
        assert abs(float(loss) - ref_l) < 2e-4 * ref_l + 2e-7, (k, i, float(loss), ref_l)
        assert rel(graw.numpy(), g["m2_grad"][k, i]) < 2e-4, (k, i, graw.numpy(), g["m2_grad"][k, i])
        crop = photo[:, :, 10:-10, 10:-10]
        assert abs(float(RC.psnr(img.detach(), crop)) - g["m2_psnr"][k, i]) < 1e-3
        assert abs(float(RC.ssim(img.detach(), crop)) - g["m2_ssim"][k, i]) < 1e-5


def test_oracle_free_running_loops_track_the_reference(golden):
    """Free-running fp32 oracle, ten iterations of each mode from start 2: the same trajectory as the reference's (the float64
    oracle's 50-iteration runs from all eight starts are in the fixture: test_fixture_is_a_converging_population)."""
    g = golden("refine50")
    k, n = 2, 10
    a = RC.refine(problem(g, torch.float32, k, 3), float(g["lr"][0]), float(g["lr"][1]), n)
    assert np.abs(a["r"].numpy() - g["m3_r"][k, :n]).max() < 2e-5 and np.abs(a["t"].numpy() - g["m3_t"][k, :n]).max() < 2e-5
    assert rel(a["losses"].numpy(), g["m3_loss"][k, :n]) < 2e-4
    b = RC.refine_apr(problem(g, torch.float32, k, 2), torch.from_numpy(g["m2_weight"][k]), torch.from_numpy(g["m2_bias"][k]),
                      photo_of(g), float(g["m2_lr"]), n)
    assert np.abs(b["poses"].numpy() - g["m2_pose"][k, :n]).max() < 2e-5
    assert rel(b["losses"].numpy(), g["m2_loss"][k, :n]) < 2e-4


# ---- the reference's own loop resolution: 60 x 80 rays (DFM_APR_refine.py:107, seven_scenes_colmap.py:264-276), one start x 50
# ---- iterations x both modes (tests/golden/refine50_60x80.npz; `tools/make_golden_refine50.py --hw 240 320 300 --k 1`) ----------
def test_oracle_matches_reference_at_60x80(golden):
    g = golden("refine50_60x80")
    assert tuple(int(v) // int(g["tinyscale"]) for v in g["hwf"][:2]) == (60, 80) and len(g["init_c2w"]) == 1
    p = problem(g, torch.float32, 0, 3)
    for i in (0, 49):
        r0 = np.zeros(3, np.float32) if i == 0 else g["m3_r"][0, i - 1]
        t0 = np.zeros(3, np.float32) if i == 0 else g["m3_t"][0, i - 1]
        loss, grad = p.loss_and_grad(r0, t0)
        ref_l = float(g["m3_loss"][0, i])
        assert abs(float(loss) - ref_l) < 2e-4 * ref_l + 2e-7, (i, float(loss), ref_l)
        assert rel(grad.numpy(), g["m3_grad"][0, i]) < 5e-4, (i, grad.numpy(), g["m3_grad"][0, i])
    # the loop converges at this resolution too, in both modes, and the float64 oracle ends where the reference ends
    for tag, key in (("m3", "m3_pose"), ("m2", "m2_final")):
        assert g[tag + "_err"][0, 0] < g["init_err"][0, 0] / 3
        e64 = RC.pose_error(g["true_c2w"], g[key + "_f64"][0])
        assert abs(e64[0] - g[tag + "_err"][0, 0]) < 0.01 * g[tag + "_err"][0, 0], (tag, e64, g[tag + "_err"][0])
