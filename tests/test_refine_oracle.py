"""The refinement-loop oracle (oracle/refine_cpu.py) and the product's torch-side pieces of the loop against vectors the
REFERENCE's own functions produced on the CPU (tools/make_golden_refine.py -> tests/golden/refine.npz, affine.npz):
LearnPose -> fix_coord_supp -> render -> affine_color_transform -> run_fusion_net -> feature_loss -> Adam, twelve iterations
of script/dm/DFM_pose_refine.py:290-348 on a 12x16-ray frame (BASELINE configs[4]/[5] in miniature).  CPU only."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from oracle import refine_cpu as RC


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def problem(g, dtype):
    from nefes_amd.field import NeRFH_NFF
    Wd, C = int(g["Wd"]), int(g["C"])
    net = NeRFH_NFF('coarse', W=Wd, f_dim=C)                       # seed-0 init == the reference's (tests/test_pose.py checksums)
    fsd = {k: v.detach().clone() for k, v in net.fusion_net.state_dict().items()}
    cfg = O.RenderCfg()
    cfg.N_samples, cfg.N_importance = int(g["Nc"]), int(g["Ni"])
    world = dict(pose_scale=float(g["pose_scale"]), pose_scale2=float(g["pose_scale2"]), move_all_cam_vec=g["move_all_cam_vec"].tolist())
    return RC.Problem(O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C), fsd,
                      torch.from_numpy(g["exposure_params"]), cfg, g["hwf"].tolist(), int(g["tinyscale"]), float(g["near"]),
                      float(g["far"]), torch.from_numpy(g["init_c2w"]), torch.from_numpy(g["target"]), torch.from_numpy(g["hist"]),
                      world, dtype=dtype)


def test_affine_color_transform_matches_reference(golden):
    """nerfh_nff.py:605-626 executed by the reference (exposure network = the fp32 stand-in, UNPINNED): the oracle's and the
    product's restatements give the same colours."""
    from nefes_amd.field import NeRFH_NFF
    g = golden("affine")
    params, hist, rgb = (torch.from_numpy(g[k]) for k in ("exposure_params", "hist", "rgb_in"))
    out = RC.affine_color_transform(params, rgb, hist, 2)
    assert rel(out.numpy(), g["rgb_out"]) < 1e-6
    assert rel(RC.exposure_mlp(params, hist.long()).numpy(), g["a_embedded"]) < 1e-6
    net = NeRFH_NFF('coarse', W=128, f_dim=16)
    with torch.no_grad():
        net.exposure_embedding.params.copy_(params)
        mine = net.affine_color_transform(types.SimpleNamespace(encode_hist=True), rgb.clone(), hist, 2)
    assert rel(mine.numpy(), g["rgb_out"]) < 1e-6
    assert rel(net.a_embedded.numpy(), g["a_embedded"]) < 1e-6


def test_adam_restatement_reproduces_the_reference_trajectory(golden):
    """Fed the reference's own gradient sequence, oracle Adam walks the reference's (r, t)."""
    g = golden("refine")
    r, t = torch.zeros(3), torch.zeros(3)
    opt = RC.Adam([r, t], [float(g["lr"][0]), float(g["lr"][1])])
    for i in range(len(g["losses"])):
        gi = torch.from_numpy(g["grads"][i])
        opt.step([gi[:3], gi[3:]])
        assert np.abs(r.numpy() - g["r"][i]).max() < 2e-7 and np.abs(t.numpy() - g["t"][i]).max() < 2e-7, i


def test_oracle_iteration_matches_reference_at_every_pose_of_the_trajectory(golden):
    """Teacher-forced: at the pose the reference held before each of its iterations, the oracle (fp32) returns the
    reference's loss and its gradient to (r, t).  This is the per-iteration parity of the whole chain; Adam on top of it is
    the test above."""
    g = golden("refine")
    p = problem(g, torch.float32)
    for i in (0, 1, 4, 8, 11):
        r0 = np.zeros(3, np.float32) if i == 0 else g["r"][i - 1]
        t0 = np.zeros(3, np.float32) if i == 0 else g["t"][i - 1]
        loss, grad = p.loss_and_grad(r0, t0)
        assert abs(float(loss) - float(g["losses"][i])) < 2e-4 * float(g["losses"][i]) + 2e-7, (i, float(loss), float(g["losses"][i]))
        assert rel(grad.numpy(), g["grads"][i]) < 1e-4, (i, grad.numpy(), g["grads"][i])


def test_oracle_free_running_loop_tracks_the_reference(golden):
    """The free-running loop: op-exact for the first iterations, then the fp32 programs drift apart the way ANY two roundings
    of this loop do.  Adam divides each gradient component by its own running magnitude, and the translation gradient of this
    scene is a small sum of cancelling terms: its rounding noise is amplified by two orders of magnitude per iteration
    (1e-9 -> 4e-9 -> 1.5e-6 -> 9e-5 in t).  The float64 run measures that conditioning: the REFERENCE's fp32 translation ends
    0.03 away from the float64 one (having moved 0.07); the rotation parameters and the loss curve are well conditioned and
    are what the bounds below pin: r within 1e-3 (it moves by 0.11), loss curve within 1e-3 of its largest value."""
    g = golden("refine")
    n = len(g["losses"])
    a = RC.refine(problem(g, torch.float32), float(g["lr"][0]), float(g["lr"][1]), n)
    b = RC.refine(problem(g, torch.float64), float(g["lr"][0]), float(g["lr"][1]), n)
    for k in ("r", "t"):
        assert np.abs(a[k][:2].numpy() - g[k][:2]).max() < 1e-6, k            # two iterations: the same trajectory
    assert rel(a["grads"][:2].numpy(), g["grads"][:2]) < 1e-5
    assert rel(a["losses"].numpy(), g["losses"]) < 1e-3 and rel(b["losses"].numpy(), g["losses"]) < 1e-3
    assert np.abs(a["r"].numpy() - g["r"]).max() < 1e-3 and np.abs(b["r"].numpy() - g["r"]).max() < 1e-3
    e_ref_t = np.abs(b["t"].numpy() - g["t"]).max()                            # reference fp32 vs float64: the loop's own noise
    assert np.abs(a["t"].numpy() - g["t"]).max() < max(2e-3, 1.5 * e_ref_t)
    assert np.abs(g["r"][-1]).max() > 0.1                                      # the motion those 1e-3 are measured against
    # the reference improves the loss it optimises by > 5x over the twelve iterations, and so does the float64 run
    assert g["losses"][-1] < 0.2 * g["losses"][0] and float(b["losses"][-1]) < 0.2 * float(b["losses"][0])
