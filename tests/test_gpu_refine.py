"""BASELINE configs[4]/[5] in miniature on the GPU: `PoseRefiner` -- LearnPose -> fix_coord_supp -> HIP
render -> affine_color_transform -> FusionNet -> feature loss -> Adam -- against tests/golden/refine.npz, twelve iterations of
script/dm/DFM_pose_refine.py:290-348 that the REFERENCE's own functions ran on the CPU (tools/make_golden_refine.py), and
against the float64 oracle loop (oracle/refine_cpu.py) for the three-way rule of tests/parity_log.py."""
import types

import numpy as np
import pytest
import torch

from oracle import refine_cpu as RC
from tests import parity_log as P
from tests.test_refine_oracle import problem, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))


def refiner(g, graph):
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.refine import PoseRefiner
    Wd, C = int(g["Wd"]), int(g["C"])
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    with torch.no_grad():
        coarse.exposure_embedding.params.copy_(T(g["exposure_params"]))
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, encode_hist=True)
    kw = dict(network_query_fn=None, perturb=0., N_importance=int(g["Ni"]), N_samples=int(g["Nc"]), network_fn=coarse,
              network_fine=fine, use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False,
              lindisp=False)
    world = dict(pose_scale=float(g["pose_scale"]), pose_scale2=float(g["pose_scale2"]), move_all_cam_vec=g["move_all_cam_vec"].tolist())
    H, W, focal = g["hwf"].tolist()
    return PoseRefiner(kw, args, (H, W, focal), float(g["near"]), float(g["far"]), tinyscale=int(g["tinyscale"]),
                       lr_r=float(g["lr"][0]), lr_t=float(g["lr"][1]), world_setup=world, graph=graph, device=DEV)


def test_refinement_iteration_matches_reference_along_its_trajectory(golden):
    """Teacher-forced: at the (r, t) the reference held before each of its twelve iterations, one PoseRefiner iteration's
    loss and gradient against the reference's (fp32, golden) and the oracle's (float64).  Loss within 2e-4 of the
    reference's (bit-identical in most iterations); gradient within max(1e-4, 3 e_ref) of the float64 one (measured: 1e-5..1.3e-4)."""
    g = golden("refine")
    ref = refiner(g, graph=False)
    ref._reset(T(g["init_c2w"]).to(DEV), T(g["target"]).to(DEV), T(g["hist"]).to(DEV))
    p64 = problem(g, torch.float64)
    worst = 0.
    for i in range(len(g["losses"])):
        r0 = np.zeros(3, np.float32) if i == 0 else g["r"][i - 1]
        t0 = np.zeros(3, np.float32) if i == 0 else g["t"][i - 1]
        with torch.no_grad():
            ref.model.r.copy_(T(r0).reshape(1, 3))
            ref.model.t.copy_(T(t0).reshape(1, 3))
        loss = float(ref.loss_and_grad())
        grad = torch.cat([ref.model.r.grad[0], ref.model.t.grad[0]]).cpu().numpy()
        l64, g64 = p64.loss_and_grad(r0, t0)
        g64 = g64.numpy()
        e_hip, e_ref, direct = rel(grad, g64), rel(g["grads"][i], g64), rel(grad, g["grads"][i])
        # factor 3 rather than the 1.5 of the branch-pinned tests: ReLU decisions are NOT pinned to the oracle's here
        P.check(f"refine_iteration[{i}]", "d loss / d (r, t)", e_hip, e_ref, direct, factor=3.0)
        el_hip, el_ref = abs(loss - float(l64)) / float(l64), abs(float(g["losses"][i]) - float(l64)) / float(l64)
        P.check(f"refine_iteration[{i}]", "loss", el_hip, el_ref, abs(loss - float(g["losses"][i])) / float(g["losses"][i]), tol=2e-4, factor=3.0)
        worst = max(worst, direct)
    assert worst < 1e-3


@pytest.mark.parametrize("graph", [False, True])
def test_refinement_loop_tracks_the_reference(golden, graph):
    """Free-running, eager and as one replayed HIP graph per iteration.  The first iterations coincide with the reference's.
    After twelve, the well-conditioned outputs agree: rotation block within 2e-3 (it moves by 0.1), loss curve within 2e-3 of
    its largest value (measured 1e-3; the reference's own fp32 curve is 8e-4 from the float64 one).  The translation is driven by a noise-level gradient that Adam amplifies (tests/test_refine_oracle.py:
    the reference's own fp32 run ends 0.03 from the float64 run): it is held to 1.5 x that distance, the shared rule."""
    g = golden("refine")
    n = len(g["losses"])
    ref = refiner(g, graph=graph)
    pose, losses = ref.refine(T(g["init_c2w"]), T(g["target"]), T(g["hist"]), n)
    losses, pose = losses.cpu().numpy(), pose[:3, :4].cpu().numpy()
    tag = f"refine_loop[{'graph' if graph else 'eager'}]"
    b = RC.refine(problem(g, torch.float64), float(g["lr"][0]), float(g["lr"][1]), n)
    truth, gold = b["poses"][-1].numpy(), g["poses"][-1]
    P.check(tag, "refined rotation block (abs)", float(np.abs(pose[:, :3] - truth[:, :3]).max()),
            float(np.abs(gold[:, :3] - truth[:, :3]).max()), float(np.abs(pose[:, :3] - gold[:, :3]).max()), tol=2e-3, factor=1.5)
    P.check(tag, "refined translation (abs)", float(np.abs(pose[:, 3] - truth[:, 3]).max()),
            float(np.abs(gold[:, 3] - truth[:, 3]).max()), float(np.abs(pose[:, 3] - gold[:, 3]).max()), tol=2e-3, factor=1.5)
    assert rel(losses[:2], g["losses"][:2]) < 2e-4
    el = rel(losses, g["losses"])
    P.record(tag, "loss curve", e_hip=rel(losses, b["losses"].numpy()), e_ref=rel(g["losses"], b["losses"].numpy()), direct=el, bound=2e-3)
    assert el < 2e-3
    assert losses[-1] < 0.2 * losses[0]
    assert float(np.abs(gold[:, :3] - g["init_c2w"][:3, :3]).max()) > 0.05          # the motion the 2e-3 are measured against


def test_fusion_net_and_affine_transform_on_the_gpu(golden):
    """nerfh_nff.py:356-418,578-626 on the device (MIOpen convolutions, BatchNorm in train mode) against the reference's CPU run."""
    from nefes_amd.field import NeRFH_NFF
    g = golden("fusion")
    net = NeRFH_NFF('coarse', W=128, f_dim=16).to(DEV)
    _, r_feat, fused = net.run_fusion_net(T(g["rgb"]).to(DEV).clone(), T(g["feat"]).to(DEV).clone(), 6, 8, 1)
    e = rel(fused.detach().cpu().numpy(), g["fused"])
    P.record("fusion_net_gpu", "fused features", e_hip=e, e_ref=None, direct=e, bound=1e-5)
    assert e < 1e-5 and np.array_equal(r_feat.cpu().numpy(), g["render_feat"])
    a = golden("affine")
    with torch.no_grad():
        net.exposure_embedding.params.copy_(T(a["exposure_params"]))
        out = net.affine_color_transform(types.SimpleNamespace(encode_hist=True), T(a["rgb_in"]).to(DEV), T(a["hist"]).to(DEV), 2)
    e = rel(out.cpu().numpy(), a["rgb_out"])
    P.record("affine_color_transform_gpu", "rgb", e_hip=e, e_ref=None, direct=e, bound=1e-6)
    assert e < 1e-6
