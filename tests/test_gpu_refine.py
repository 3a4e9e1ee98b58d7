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
from tests.test_refine_oracle import problem, rel  # noqa: E402

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


def test_refinement_loop_graph_equals_eager_and_follows_the_loss_curve(golden):
    """Free-running on the round-2 fixture (default-init scene), eager and as one replayed HIP graph per iteration: the two are the
    same arithmetic (poses and loss curves bit for bit), the first iterations coincide with the reference's, and the loss curve obeys
    the shared rule against the float64 oracle loop.  This scene's translation is driven by a noise-level gradient that Adam
    amplifies (the reference's own fp32 run ends 0.03 from the float64 run: tests/test_refine_oracle.py), so its refined POSE is not
    a parity statement and is not asserted here (round 3 held it to 1.5 x that distance, a 0.047 bound that proved nothing); the
    free-running pose statements live on the conditioned scene of tests/test_gpu_refine50.py."""
    g = golden("refine")
    n = len(g["losses"])
    out = {}
    for graph in (False, True):
        ref = refiner(g, graph=graph)
        pose, losses = ref.refine(T(g["init_c2w"]), T(g["target"]), T(g["hist"]), n)
        out[graph] = (pose[:3, :4].cpu().numpy(), losses.cpu().numpy())
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    pose, losses = out[True]
    b = RC.refine(problem(g, torch.float64), float(g["lr"][0]), float(g["lr"][1]), n)
    assert rel(losses[:2], g["losses"][:2]) < 2e-4
    P.check("refine_loop", "loss curve", rel(losses, b["losses"].numpy()), rel(g["losses"], b["losses"].numpy()), rel(losses, g["losses"]),
            tol=1e-3, factor=1.5)
    assert losses[-1] < 0.2 * losses[0]


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


# ---- the loop's glue as library kernels (csrc/refine.hip, csrc/upsample.hip) against the torch expressions they replace ----
@pytest.mark.parametrize("r0", [(0., 0., 0.), (0.11, -0.07, 0.03), (1e-6, -2e-6, 5e-7), (1.2, 0.7, -2.1)])
def test_pose_compose_matches_learnpose_and_fix_coord(r0):
    """nefes_pose_compose_fwd/bwd == LearnPose.forward (poses.py:43-50) + fix_coord_supp (direct_pose_model.py:224-231) and
    their autograd, evaluated by torch in float64 (the kernel computes in float64 and rounds once)."""
    from nefes_amd import ops
    from nefes_amd.pose import LearnPose
    from nefes_amd.refine import fix_coord_supp
    g = torch.Generator().manual_seed(3)
    init = torch.eye(4, dtype=torch.float64)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    init[:3, :3], init[:3, 3] = q, torch.randn(3, generator=g, dtype=torch.float64)
    ws = dict(pose_scale=0.8, pose_scale2=1.25, move_all_cam_vec=[0.1, -0.05, 0.2])
    G = torch.randn(3, 4, generator=g, dtype=torch.float64)
    m = LearnPose(1, True, True, init_c2w=init[None].clone()).double()
    with torch.no_grad():
        m.r.copy_(torch.tensor([r0], dtype=torch.float64))
        m.t.copy_(torch.tensor([[0.3, -0.2, 0.1]], dtype=torch.float64))
    ref = fix_coord_supp(m(0)[None, :3, :4], ws)[0]
    (ref * G).sum().backward()
    r = m.r.detach().float().reshape(3).to(DEV).requires_grad_()
    t = m.t.detach().float().reshape(3).to(DEV).requires_grad_()
    out = ops.pose_compose(r, t, init.float().to(DEV), ws["pose_scale"], ws["move_all_cam_vec"], ws["pose_scale2"])
    (out * G.float().to(DEV)).sum().backward()
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 3e-7
    assert rel(r.grad.cpu().numpy(), m.r.grad[0].numpy()) < 1e-6 and rel(t.grad.cpu().numpy(), m.t.grad[0].numpy()) < 1e-6


def test_svd_reg_kernels_match_float64_svd():
    """ops.svd_reg (nefes_svd_reg_fwd/bwd: one-sided Jacobi in float64, the polar factor's own derivative) == svd_reg
    (dm/DFM_pose_refine.py:119-129) through torch.svd + autograd in float64: near-rotations (what a pose network regresses; where
    autograd through the fp32 SVD loses four digits), general matrices, det < 0, a scaled rotation, a batch; the translation column
    and its gradient pass through.  The fp32 torch path's own error is recorded beside it."""
    from nefes_amd import ops
    g = torch.Generator().manual_seed(12)
    q, _ = torch.linalg.qr(torch.randn(6, 3, 3, generator=g, dtype=torch.float64))
    A = torch.cat([q + 1e-3 * torch.randn(6, 3, 3, generator=g, dtype=torch.float64),          # near-rotations
                   torch.randn(4, 3, 3, generator=g, dtype=torch.float64),                      # general
                   2.5 * q[:2], torch.diag(torch.tensor([1., 1., -1.], dtype=torch.float64))[None] @ q[:1]])
    pose = torch.cat([A, torch.randn(A.shape[0], 3, 1, generator=g, dtype=torch.float64)], -1).float()
    G = torch.randn(pose.shape, generator=g, dtype=torch.float64)

    def torch_svd_reg(p):
        u, _, v = torch.svd(p[..., :3, :3])
        return torch.cat([u @ v.transpose(-2, -1), p[..., :3, 3:]], -1)
    pd = pose.double().requires_grad_()
    rd = torch_svd_reg(pd)
    (rd * G).sum().backward()
    ph = pose.to(DEV).requires_grad_()
    rh = ops.svd_reg(ph)
    (rh * G.float().to(DEV)).sum().backward()
    pt = pose.to(DEV).requires_grad_()
    rt = torch_svd_reg(pt)
    (rt * G.float().to(DEV)).sum().backward()
    e_val, e_grad = rel(rh.detach().cpu().numpy(), rd.detach().numpy()), rel(ph.grad.cpu().numpy(), pd.grad.numpy())
    e_val_t, e_grad_t = rel(rt.detach().cpu().numpy(), rd.detach().numpy()), rel(pt.grad.cpu().numpy(), pd.grad.numpy())
    P.record("svd_reg_gpu", "pose after svd_reg", e_hip=e_val, e_ref=e_val_t, direct=e_val, bound=3e-7)
    P.record("svd_reg_gpu", "d / d regressed pose", e_hip=e_grad, e_ref=e_grad_t, direct=e_grad, bound=1e-6)
    assert e_val < 3e-7 and e_grad < 1e-6, (e_val, e_grad, e_val_t, e_grad_t)
    r3 = rh.detach()[:, :3, :3].double()
    assert float((r3 @ r3.transpose(1, 2) - torch.eye(3, device=DEV, dtype=torch.float64)).abs().max()) < 5e-7
    assert torch.equal(rh.detach()[:, :, 3], ph.detach()[:, :, 3]) and torch.equal(ph.grad[:, :, 3], G.float().to(DEV)[:, :, 3])
    one = ops.svd_reg(pose[0].to(DEV))                                  # a single [3,4] pose keeps its shape
    assert one.shape == (3, 4) and torch.equal(one, rh.detach()[0])


@pytest.mark.parametrize("do_svd", [True, False])
def test_regressed_pose_kernels_equal_svd_reg_then_fix_coord_supp(do_svd):
    """ops.regressed_pose (nefes_regressed_pose_fwd/bwd: what train_on_batch does to the regression network's output before it renders,
    DFM_APR_refine.py:91-97, one launch each way) == ops.svd_reg followed by refine.fix_coord_supp's torch expression: the rotation block
    bit for bit (same kernel body), the translation bit for bit (one fp32 multiply and one add, in that order), gradients to rounding."""
    from nefes_amd import ops
    from nefes_amd.refine import fix_coord_supp
    g = torch.Generator().manual_seed(21)
    q, _ = torch.linalg.qr(torch.randn(5, 3, 3, generator=g, dtype=torch.float64))
    pose = torch.cat([q + 1e-3 * torch.randn(5, 3, 3, generator=g, dtype=torch.float64), torch.randn(5, 3, 1, generator=g, dtype=torch.float64)], -1).float()
    G = torch.randn(pose.shape, generator=g).to(DEV)
    ws = dict(pose_scale=0.3027, pose_scale2=0.83, move_all_cam_vec=[0.11, -0.23, 0.07])
    a = pose.to(DEV).requires_grad_()
    ra = ops.regressed_pose(a, do_svd, ws)
    (ra * G).sum().backward()
    b = pose.to(DEV).requires_grad_()
    rb = fix_coord_supp(ops.svd_reg(b) if do_svd else b, ws)
    (rb * G).sum().backward()
    assert torch.equal(ra.detach(), rb.detach())
    e = rel(a.grad.cpu().numpy(), b.grad.cpu().numpy())
    assert e < 2e-7, e
    # no world set-up: svd_reg alone
    c = pose.to(DEV).requires_grad_()
    rc = ops.regressed_pose(c, do_svd, None)
    assert torch.equal(rc.detach(), (ops.svd_reg(pose.to(DEV)) if do_svd else pose.to(DEV)))
    one = ops.regressed_pose(pose[0].to(DEV), do_svd, ws)
    assert one.shape == (3, 4) and torch.equal(one, ra.detach()[0])


@pytest.mark.parametrize("C,P", [(128, 220 * 300), (16, 12 * 16), (3, 1000)])
def test_cosine_feature_loss_matches_torch(C, P):
    """nefes_cosine_loss_fwd/bwd == feature_loss (DFM_pose_refine.py:211-233) in float64 torch, value and gradient; one channel is
    all zeros (the eps clamp of CosineSimilarity)."""
    from nefes_amd import ops
    from nefes_amd.refine import feature_loss
    g = torch.Generator().manual_seed(C)
    a, b = torch.randn(C, P, generator=g), torch.randn(C, P, generator=g)
    b = b + 0.7 * a                               # correlated, as rendered and target features are
    a[0] = 0.
    ad = a.double().requires_grad_()
    ref = feature_loss(ad.reshape(C, 1, P), b.double().reshape(C, 1, P))
    ref.backward()
    ag = a.to(DEV).requires_grad_()
    out = ops.cosine_feature_loss(ag.reshape(C, 1, P), b.to(DEV).reshape(C, 1, P))
    (3.0 * out).backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 2e-7
    assert rel(ag.grad.cpu().numpy() / 3.0, ad.grad.numpy()) < 1e-6


def test_bicubic_upsample_window_equals_cropped_full_image():
    """The windowed up-sampling (crop=10) == the full up-sampling sliced [10:-10, 10:-10], forward and backward."""
    from nefes_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 5, 15, 20, generator=g).to(DEV)
    go = torch.randn(1, 5, 40, 60, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = ops.bicubic_upsample(xa, (60, 80), crop=10)
    yb = ops.bicubic_upsample(xb, (60, 80))[:, :, 10:-10, 10:-10]
    ya.backward(go)
    yb.backward(go)
    assert ya.shape == yb.shape == (1, 5, 40, 60) and torch.equal(ya, yb)
    assert torch.allclose(xa.grad, xb.grad, rtol=0, atol=2e-6 * float(xb.grad.abs().max()))


@pytest.mark.parametrize("C,h,w,ts", [(128, 60, 80, 4), (5, 15, 20, 4), (3, 30, 40, 4)])
def test_upsampled_cosine_loss_equals_separate_kernels(C, h, w, ts):
    """ops.upsampled_cosine_loss (up-sampling, 10 px crop and feature loss in one pass each way; the up-sampled image is never
    written) == ops.cosine_feature_loss(ops.bicubic_upsample(x, crop=10), target): the same loss to 1e-7 (same interpolation
    expression, float64 sums in another order), the same gradient to 1e-6 (the gather runs along x first instead of y first), and
    both against torch's own Upsample + CosineSimilarity in float64."""
    from nefes_amd import ops
    from nefes_amd.refine import feature_loss
    g = torch.Generator().manual_seed(C + h)
    H, W = h * ts, w * ts
    x = torch.randn(1, C, h, w, generator=g)
    tgt = torch.randn(C, H - 20, W - 20, generator=g) + 0.5 * torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic")[0, :, 10:-10, 10:-10]
    xa, xb = x.to(DEV).requires_grad_(), x.to(DEV).requires_grad_()
    la, cos = ops.upsampled_cosine_loss(xa, tgt.to(DEV), (H, W), crop=10, return_cos=True)
    lb = ops.cosine_feature_loss(ops.bicubic_upsample(xb, (H, W), crop=10)[0], tgt.to(DEV))
    (2.0 * la).backward()
    (2.0 * lb).backward()
    assert abs(float(la) - float(lb)) < 1e-7 and cos.shape == (C,)
    assert rel(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) < 1e-6
    xd = x.double().requires_grad_()
    ld = feature_loss(torch.nn.functional.interpolate(xd, size=(H, W), mode="bicubic")[0, :, 10:-10, 10:-10], tgt.double())
    (2.0 * ld).backward()
    assert abs(float(la) - float(ld)) < 3e-7 and rel(xa.grad.cpu().numpy(), xd.grad.numpy()) < 2e-6


@pytest.mark.parametrize("C,h,w,OH,OW,crop", [(128, 60, 80, 240, 320, 10), (5, 15, 20, 60, 80, 10), (3, 9, 11, 31, 45, 3), (2, 8, 8, 32, 32, 0),
                                              (4, 6, 70, 24, 280, 1)])
def test_prepared_target_loss_equals_one_pass_kernels_and_torch(C, h, w, OH, OW, crop):
    """ops.upsampled_cosine_loss_prepared (the fixed target folded through the up-sampling once: <up, t> = <x, Uy^T t Ux>,
    |up|^2 = <x, Gy x Gx>; csrc/refine.hip upcos_gram_*) == ops.upsampled_cosine_loss == torch's Upsample + CosineSimilarity in
    float64 (DFM_APR_refine.py:114-131), value, per-channel similarities and gradient; integer and non-integer scales, with and
    without a crop, one all-zero channel (the eps clamp), and a second target through the same buffers (what a captured graph reads)."""
    from nefes_amd import ops
    from nefes_amd.refine import feature_loss
    g = torch.Generator().manual_seed(C * h + OW)
    x = torch.randn(1, C, h, w, generator=g)
    x[0, C - 1] = 0.
    up = torch.nn.functional.interpolate(x, size=(OH, OW), mode="bicubic")[0, :, crop:OH - crop, crop:OW - crop]
    prep = ops.UpcosTarget(C, h, w, OH, OW, crop, DEV)
    assert prep.band <= 3 and float((prep.gx - prep.gx.T).abs().max()) == 0.0
    for k in range(2):
        tgt = torch.randn(C, OH - 2 * crop, OW - 2 * crop, generator=g) + 0.5 * up
        prep.update(tgt.to(DEV))
        xa, xb = x.to(DEV).requires_grad_(), x.to(DEV).requires_grad_()
        la, cos_a = ops.upsampled_cosine_loss_prepared(xa, prep, return_cos=True)
        lb, cos_b = ops.upsampled_cosine_loss(xb, tgt.to(DEV), (OH, OW), crop=crop, return_cos=True)
        (2.0 * la).backward()
        (2.0 * lb).backward()
        xd = x.double().requires_grad_()
        ld = feature_loss(torch.nn.functional.interpolate(xd, size=(OH, OW), mode="bicubic")[0, :, crop:OH - crop, crop:OW - crop], tgt.double())
        (2.0 * ld).backward()
        e_prep, e_pass = rel(xa.grad.cpu().numpy(), xd.grad.numpy()), rel(xb.grad.cpu().numpy(), xd.grad.numpy())
        P.record(f"upcos_prepared[{C},{h},{w},{OH},{OW},{crop}][{k}]", "d loss / d fused features (vs float64 torch)", e_hip=e_prep, e_ref=None,
                 direct=e_prep, bound=2e-6, one_pass_kernels=e_pass, loss_err=abs(float(la) - float(ld)), loss_err_one_pass=abs(float(lb) - float(ld)))
        assert abs(float(la) - float(lb)) < 1e-7 and abs(float(la) - float(ld)) < 3e-7
        assert float((cos_a - cos_b).abs().max()) < 1e-6 and cos_a.shape == (C,)
        assert rel(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) < 1e-6 and e_prep < 2e-6
        assert float(xa.grad[0, C - 1].abs().max()) > 0          # the clamped-norm channel still has its b / (eps |b|) gradient


def test_fused_glue_iteration_equals_torch_glue(golden):
    """One PoseRefiner iteration with the glue kernels == the same iteration with the torch expressions (APR variant: bicubic
    up-sampling + 10 px crop; DFM variant without): loss and (r, t) gradient."""
    g = golden("refine")
    H, W, _ = g["hwf"].tolist()
    for up in (False, True):
        res = []
        for fused in (True, False):
            ref = refiner(g, graph=False)
            ref.fused_glue, ref.upsample = fused, up
            tgt = T(g["target"]).to(DEV)
            if up:
                from nefes_amd import ops
                tgt = ops.bicubic_upsample(tgt[None], (int(H), int(W)), crop=10)[0]
                ref.target = torch.zeros_like(tgt)
            ref._reset(T(g["init_c2w"]).to(DEV), tgt, T(g["hist"]).to(DEV))
            with torch.no_grad():
                ref.model.r.copy_(T(g["r"][3]).reshape(1, 3))
                ref.model.t.copy_(T(g["t"][3]).reshape(1, 3))
            loss = float(ref.loss_and_grad())
            res.append((loss, torch.cat([ref.model.r.grad[0], ref.model.t.grad[0]]).cpu().numpy()))
        (la, ga), (lb, gb) = res
        assert abs(la - lb) < 2e-4 * abs(lb) + 1e-7, (up, la, lb)
        assert rel(ga, gb) < 2e-4, (up, ga, gb)


@pytest.mark.parametrize("graph", [False, True])
def test_batched_refinement_equals_one_image_at_a_time(golden, graph):
    """PoseRefiner(images=3): three query images (different initial poses, targets, histograms) refined side by side walk the
    trajectories they walk alone -- the batch shares launches, not arithmetic (per-image FusionNet normalisation, per-image
    losses, element-wise Adam).  The batched convolutions may take another MIOpen solver than the single-image ones, so the two
    runs are two fp32 roundings of the loop: equal to 2e-5 after two iterations, then drifting as any two do (test_refine_oracle.py:
    x100 per iteration in the translation) -- bounded at 2e-3 after four."""
    g = golden("refine")
    B, n = 3, 4
    gen = torch.Generator().manual_seed(11)
    inits = T(g["init_c2w"])[None].repeat(B, 1, 1).clone()
    inits[1, :3, 3] += torch.tensor([0.05, -0.02, 0.03])
    inits[2, :3, 3] += torch.tensor([-0.04, 0.06, -0.01])
    targets = T(g["target"])[None].repeat(B, 1, 1, 1).clone()
    targets[1] += 0.05 * torch.randn(targets[1].shape, generator=gen)
    targets[2] += 0.10 * torch.randn(targets[2].shape, generator=gen)
    hists = torch.stack([T(g["hist"])[0], T(g["hist"])[0].flip(0), torch.full((10,), 10.)])
    single = refiner(g, graph=graph)
    alone = [single.refine(inits[b], targets[b], hists[b][None], n) for b in range(B)]
    from nefes_amd.refine import PoseRefiner
    batch = PoseRefiner(single.kw, single.args, (single.H, single.W, single.focal * (single.H // single.h)), single.near, single.far,
                        tinyscale=single.H // single.h, lr_r=float(g["lr"][0]), lr_t=float(g["lr"][1]), world_setup=single.world_setup,
                        graph=graph, device=DEV, images=B)
    poses, losses = batch.refine(inits, targets, hists, n)
    assert poses.shape == (B, 4, 4) and losses.shape == (n, B)
    for b in range(B):
        p1, l1 = alone[b]
        assert rel(losses[:, b].cpu().numpy(), l1.cpu().numpy()) < 2e-4, (b, losses[:, b], l1)
        assert float((poses[b] - p1).abs().max()) < 2e-3, (b, (poses[b] - p1).abs().max())
    poses2, _ = batch.refine(inits, targets, hists, 2)                  # re-uses the captured graph
    for b in range(B):
        p1, _ = single.refine(inits[b], targets[b], hists[b][None], 2)
        assert float((poses2[b] - p1).abs().max()) < 2e-5, (b, (poses2[b] - p1).abs().max())


@pytest.mark.parametrize("upsample,encode_hist,world", [(False, False, False), (True, True, True), (True, False, False)])
def test_pose_refiner_option_matrix(golden, upsample, encode_hist, world):
    """PoseRefiner(images=2) across its options (APR / DFM variant, with and without the exposure transform and the world
    set-up): finite, decreasing losses, and the fused glue equals the torch glue for one image."""
    from nefes_amd import ops
    from nefes_amd.refine import PoseRefiner
    g = golden("refine")
    base = refiner(g, graph=False)
    args = types.SimpleNamespace(**vars(base.args))
    args.encode_hist = encode_hist
    H, W = base.H, base.W
    ws = base.world_setup if world else None
    tgt = T(g["target"]).to(DEV)
    if upsample:
        tgt = ops.bicubic_upsample(tgt[None], (H, W), crop=10)[0]
    mk = lambda **k: PoseRefiner(base.kw, args, (H, W, base.focal * (H // base.h)), base.near, base.far, tinyscale=H // base.h,
                                 lr_r=0.01, lr_t=0.01, world_setup=ws, upsample=upsample, graph=False, device=DEV, **k)
    init, hist = T(g["init_c2w"]).to(DEV), T(g["hist"]).to(DEV)
    p1, l1 = mk().refine(init, tgt, hist, 3)
    p0, l0 = mk(fused_glue=False).refine(init, tgt, hist, 3)
    assert torch.isfinite(l1).all() and rel(l1.cpu().numpy(), l0.cpu().numpy()) < 5e-4 and float((p1 - p0).abs().max()) < 5e-4
    p2, l2 = mk(images=2).refine(init[None].repeat(2, 1, 1), tgt[None].repeat(2, 1, 1, 1), hist.repeat(2, 1), 3)
    assert l2.shape == (3, 2) and torch.isfinite(l2).all() and float(l2[-1].max()) < float(l2[0].min())
    assert rel(l2[:, 0].cpu().numpy(), l1.cpu().numpy()) < 5e-4 and rel(l2[:, 1].cpu().numpy(), l1.cpu().numpy()) < 5e-4


@pytest.mark.parametrize("B,Cin,Cout,k,relu,H,W", [(1, 131, 64, 3, True, 60, 80), (1, 64, 64, 3, True, 60, 80), (1, 64, 128, 5, False, 60, 80),
                                                   (3, 19, 64, 3, True, 17, 23), (2, 64, 16, 5, False, 9, 70),
                                                   (1, 1, 1, 5, True, 3, 3), (3, 9, 33, 5, True, 7, 11), (2, 7, 5, 3, False, 13, 9)])
def test_frozen_conv_matches_float64(B, Cin, Cout, k, relu, H, W):
    """csrc/conv.hip (ops.frozen_conv2d): Conv2d(stride 1, same padding)[+ReLU] forward and input gradient against torch's
    convolution in float64; torch's own fp32 GPU convolution (MIOpen) measured beside it.  Ragged sizes: odd channel counts,
    channel counts that are no multiple of 32, a pixel count that is no multiple of 32, images narrower than a pixel tile; a single
    channel, fewer channel pairs than the four waves that split them, an image smaller than the 5x5 kernel."""
    from nefes_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    gy = torch.randn(B, Cout, H, W, generator=g)

    def ref(dt, dev):
        xx = x.to(dev, dt).requires_grad_()
        y = torch.nn.functional.conv2d(xx, w.to(dev, dt), b.to(dev, dt), padding=k // 2)
        y = torch.relu(y) if relu else y
        (y * gy.to(dev, dt)).sum().backward()
        return y.detach().cpu().double(), xx.grad.cpu().double()
    y64, g64 = ref(torch.float64, "cpu")
    y32, g32 = ref(torch.float32, DEV)
    xd = x.to(DEV).requires_grad_()
    y = ops.frozen_conv2d(xd, w.to(DEV), b.to(DEV), relu=relu)
    (y * gy.to(DEV)).sum().backward()
    rel = lambda a_, t_: float((a_.cpu().double() - t_).abs().max() / t_.abs().max())
    e_y, e_g, r_y, r_g = rel(y.detach(), y64), rel(xd.grad, g64), rel(y32, y64), rel(g32, g64)
    P.record(f"frozen_conv[{B},{Cin},{Cout},{k}]", "output / input gradient vs float64", e_hip=max(e_y, e_g), e_ref=max(r_y, r_g), bound=1e-5)
    assert e_y < 1e-5 and e_g < 1e-5, (e_y, e_g, r_y, r_g)


def test_fusion_net_hip_convs_against_float64():
    """FusionNet.forward_parts with frozen weights takes the HIP convolutions: fused features and the gradient to the render
    against the same module in float64 on the CPU, with torch's fp32 GPU layers (MIOpen) measured beside it.  One image
    (BatchNorm on its statistics) and three images (instance norm).  (MIOpen's fp32 backward-data kernels for this shape are
    ~5e-3 away from float64; the implicit-GEMM kernels ~1e-6.)"""
    import copy
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    torch.manual_seed(4)
    net = NeRFH_NFF('coarse', W=128, f_dim=128).requires_grad_(False).to(DEV)
    net64 = copy.deepcopy(net).cpu().double()
    relm = lambda a_, t_: float((a_.cpu().double() - t_).abs().max() / t_.abs().max())
    for B in (1, 3):
        rgb = torch.rand(B * 4800, 3, device=DEV)
        feat = torch.randn(B * 4800, 128, device=DEV)
        wl = torch.randn(B, 128, 60, 80, device=DEV)
        r, f = rgb.cpu().double().requires_grad_(), feat.cpu().double().requires_grad_()
        _, _, fused = net64.run_fusion_net(r, f, 60, 80, B, per_image_norm=B > 1)
        (fused * wl.cpu().double()).sum().backward()
        truth = (fused.detach(), r.grad, f.grad)
        err = {}
        for hip in (True, False):
            net.fusion_net.HIP_CONVS = hip
            r, f = rgb.clone().requires_grad_(), feat.clone().requires_grad_()
            ops.TIMERS = {}
            _, _, fused = net.run_fusion_net(r, f, 60, 80, B, per_image_norm=B > 1)
            assert ("conv2d_same" in ops.TIMERS) == hip
            ops.TIMERS = None
            (fused * wl).sum().backward()
            err[hip] = [relm(a_, t_) for a_, t_ in zip((fused.detach(), r.grad, f.grad), truth)]
        net.fusion_net.HIP_CONVS = True
        for name, e_hip, e_ref in zip(("fused", "d rgb", "d feature"), err[True], err[False]):
            P.record(f"fusion_net[{B}]", name + " vs float64 (e_ref = torch fp32 on the GPU)", e_hip=e_hip, e_ref=e_ref, bound=1e-4)
            assert e_hip < 1e-4, (B, name, e_hip, e_ref)


@pytest.mark.parametrize("B,C,H,W,per_image", [(1, 128, 60, 80, False), (3, 7, 5, 9, False), (3, 7, 5, 9, True), (8, 128, 15, 20, True)])
def test_batch_norm_train_kernels_match_torch_in_float64(B, C, H, W, per_image):
    """ops.batch_norm_train_frozen (nefes_bn_train_fwd/bwd) == torch.nn.BatchNorm2d in train mode (== instance_norm with the module's
    affine parameters for `per_image`, FusionNet.forward_parts' rule for a batch of images) in float64: output, gradient to the input,
    running mean / variance (unbiased) and the batch counter after two calls."""
    from nefes_amd import ops
    g = torch.Generator().manual_seed(B * C + H)
    x = torch.randn(B, C, H, W, generator=g) * 2.0 + 0.3
    G = torch.randn(B, C, H, W, generator=g)
    bn64 = torch.nn.BatchNorm2d(C).double().train()
    with torch.no_grad():
        bn64.weight.copy_(torch.randn(C, generator=g).double())
        bn64.bias.copy_(torch.randn(C, generator=g).double())
    import copy
    bn = copy.deepcopy(bn64).float().to(DEV).requires_grad_(False).train()
    for rep in range(2):
        xd = x.double().requires_grad_()
        if per_image:
            yd = torch.nn.functional.instance_norm(xd, weight=bn64.weight, bias=bn64.bias, eps=bn64.eps)
        else:
            yd = bn64(xd)
        (yd * G.double()).sum().backward()
        xh = x.to(DEV).requires_grad_()
        yh = ops.batch_norm_train_frozen(xh, bn, per_image)
        (yh * G.to(DEV)).sum().backward()
        e_y, e_g = rel(yh.detach().cpu().numpy(), yd.detach().numpy()), rel(xh.grad.cpu().numpy(), xd.grad.numpy())
        assert e_y < 1e-6 and e_g < 2e-6, (rep, e_y, e_g)
    P.record(f"batch_norm_train[{B},{C},{H},{W},{int(per_image)}]", "output / d input vs float64 torch", e_hip=max(e_y, e_g), e_ref=None, direct=max(e_y, e_g), bound=2e-6)
    if per_image:
        assert int(bn.num_batches_tracked) == 0 and float(bn.running_mean.abs().max()) == 0.0            # untouched
    else:
        assert int(bn.num_batches_tracked) == 2
        assert rel(bn.running_mean.cpu().numpy(), bn64.running_mean.numpy()) < 1e-6 and rel(bn.running_var.cpu().numpy(), bn64.running_var.numpy()) < 1e-6


@pytest.mark.parametrize("C,H,W,crop", [(3, 240, 320, 10), (3, 60, 80, 0), (1, 4, 5, 0), (2, 9, 33, 1)])
def test_psnr_ssim_kernels_equal_the_verification_steps_torch_expressions(C, H, W, crop):
    """ops.psnr_ssim (csrc/refine.hip psnr_ssim_*: the verification step of train_on_batch, DFM_APR_refine.py:117-128, :146-150) against
    mse2psnr(img2mse(x, y)) and SSIM()(x, y).mean() (utils/utils.py:15-49) -- the float64 value of the same expressions and torch's own
    fp32 on the device -- on cropped VIEWS, which the kernel reads through their strides."""
    from nefes_amd import ops
    from nefes_amd.refine import ssim_map, mse2psnr, img2mse
    g = torch.Generator().manual_seed(11)
    base = torch.rand(1, C, H, W, generator=g)
    other = (base + 0.1 * torch.randn(1, C, H, W, generator=g)).clamp(0, 1)
    sl = (slice(None), slice(None), slice(crop, H - crop), slice(crop, W - crop)) if crop else (slice(None),) * 4
    x, y = base.to(DEV)[sl], other.to(DEV)[sl]
    ps, ss = ops.psnr_ssim(x[0], y[0])
    x64, y64 = base.double()[sl], other.double()[sl]
    ps64, ss64 = float(mse2psnr(img2mse(x64, y64))), float(ssim_map(x64, y64).mean())
    ps32, ss32 = float(mse2psnr(img2mse(x, y))), float(ssim_map(x, y).mean())
    assert abs(float(ps) - ps64) <= 2e-6 * abs(ps64) and abs(float(ss) - ss64) <= 2e-6, (float(ps), ps64, float(ss), ss64)
    assert abs(float(ps) - ps32) <= 5e-6 * abs(ps32) and abs(float(ss) - ss32) <= 5e-6, (float(ps), ps32, float(ss), ss32)
    if crop:                                           # the view was read in place: a contiguous copy gives the same bits
        ps_c, ss_c = ops.psnr_ssim(x[0].contiguous(), y[0].contiguous())
        assert torch.equal(ps, ps_c) and torch.equal(ss, ss_c)
    same_p, same_s = ops.psnr_ssim(x[0], x[0])         # identical images: mse 0 -> psnr +inf (torch: -10 log(0) / log(10)), ssim 1
    assert torch.isinf(same_p) and float(same_p) > 0 and abs(float(same_s) - 1.0) < 1e-6
    with pytest.raises(RuntimeError, match="dense rows"):
        ops.psnr_ssim(x[0][:, :, ::2], y[0][:, :, ::2])
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.psnr_ssim(x[0][:, :3], y[0][:, :3])        # three rows: ReflectionPad2d(3) needs four


def test_verification_step_through_the_kernels_equals_the_torch_expressions(golden):
    """PoseRefiner._verification with FUSED_VERIFICATION on / off behind the same iterations of mode 2 (the refine50 fixture's scene):
    PSNR and SSIM agree to fp32 rounding -- and with the reference's own numbers for that iteration (tests/test_gpu_refine50.py holds
    every checked iteration to those)."""
    from nefes_amd.refine import PoseRefiner
    from tests.test_gpu_refine50 import TinyAPR, refiner, target_full
    from tests.test_refine50_oracle import photo_of
    g = golden("refine50")
    ref = refiner(g, apr=TinyAPR(g["m2_weight"][0], g["m2_bias"][0]))
    ref.refine_apr(photo_of(g), target_full(g), T(g["hist"]), iters=2, verification=False)
    on = ref._verification()
    try:
        PoseRefiner.FUSED_VERIFICATION = False
        off = ref._verification()
    finally:
        PoseRefiner.FUSED_VERIFICATION = True
    assert abs(on[0] - off[0]) <= 5e-6 * abs(off[0]) and abs(on[1] - off[1]) <= 5e-6, (on, off)
    assert abs(on[0] - g["m2_psnr"][0, 1]) < 2e-2 and abs(on[1] - g["m2_ssim"][0, 1]) < 2e-4, (on, g["m2_psnr"][0, 1], g["m2_ssim"][0, 1])
