"""CPU oracle for the per-image pose-refinement loop around the render path.

TEST INFRASTRUCTURE ONLY (same rule as oracle/ref_cpu.py: imported by tests/ only; the product never imports it).

A plain-torch, dtype-generic restatement of one iteration of script/dm/DFM_pose_refine.py:290-348 (`DFM_optimization_NFF`,
paths relative to /root/reference):

    LearnPose.forward         script/models/poses.py:26-50 (lietorch=False)  +  script/utils/lie_group_helper.py:60-81
    fix_coord_supp            script/dm/direct_pose_model.py:210-232
    render                    oracle/ref_cpu.py (script/models/rendering.py:197-243)
    affine_color_transform    script/models/nerfh_nff.py:605-626   (exposure network: see below)
    run_fusion_net            script/models/nerfh_nff.py:578-603, FusionNet :356-418 (BatchNorm in TRAIN mode: the reference
                              never calls .eval() on the NeRF modules -- batch statistics of the one image, biased variance)
    feature_loss              script/dm/DFM_pose_refine.py:211-233  (1 - mean over channels of the cosine similarity over pixels)
    Adam                      torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8), restated so that it runs in float64

    train_on_batch            script/dm/DFM_APR_refine.py:84-156 (`pose_only=2`, the shipped default): the pose comes from a
                              regression network (here `tiny_apr`, a stand-in for the out-of-scope CNN) + svd_reg
                              (dm/DFM_pose_refine.py:119-129); fused features bicubically up-sampled to (H, W) (:114) and
                              cropped by 10 px (:125-126); PSNR / SSIM of the verification step (:148-150, utils/utils.py:15-49)
    compute_pose_error_SE3    script/dm/pose_model.py:75-92 == script/eval.py:34-51 (translation distance, rotation angle in degrees)

Parity status: PINNED for everything above except the exposure network, by tests/golden/refine.npz and affine.npz, which
tools/make_golden_refine.py produces by running the reference's own functions on the CPU (tests/test_refine_oracle.py), and by
tests/golden/refine50.npz (tools/make_golden_refine50.py): 50 iterations x 8 perturbed starts of the reference's own
`DFM_optimization_NFF` and `train_on_batch`, called as they are (tests/test_refine50_oracle.py).
`exposure_mlp` restates tiny-cuda-nn's FullyFusedMLP(10 -> 12, 32 neurons, 3 hidden layers, ReLU, no bias) in fp32 with the
flat parameter layout of nefes_amd.field.ExposureMLP; tiny-cuda-nn is neither vendored by the reference nor installed, so its
layout and its fp16 arithmetic are UNPINNED.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import ref_cpu as O

Tensor = torch.Tensor


def skew(v: Tensor) -> Tensor:
    """lie_group_helper.py:45-57."""
    z = torch.zeros((), dtype=v.dtype)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def learn_pose(r: Tensor, t: Tensor, init_c2w: Tensor) -> Tensor:
    """poses.py:43-50 with make_c2w (lie_group_helper.py:60-81): [Exp(r) R0 | t + t0] as 3x4."""
    K = skew(r)
    n = r.norm() + 1e-15
    R = torch.eye(3, dtype=r.dtype) + (torch.sin(n) / n) * K + ((1 - torch.cos(n)) / n ** 2) * (K @ K)
    return torch.cat([R @ init_c2w[:3, :3], (t + init_c2w[:3, 3])[:, None]], 1)


def fix_coord_supp(pose: Tensor, pose_scale: float, move_all_cam_vec, pose_scale2: float) -> Tensor:
    """direct_pose_model.py:224-231: t' = (t * sc + move) * sc2 (the reference edits a view in place; same values)."""
    mv = torch.as_tensor(move_all_cam_vec, dtype=pose.dtype)
    return torch.cat([pose[:, :3], ((pose[:, 3] * pose_scale + mv) * pose_scale2)[:, None]], 1)


EXPOSURE_SHAPES = [(32, 16), (32, 32), (32, 32), (16, 32)]


def exposure_mlp(params: Tensor, hist: Tensor) -> Tensor:
    """Stand-in for tcnn.Network (nerfh_nff.py:511-522), UNPINNED: [B,10] -> [B,12]."""
    h = F.pad(hist.to(params.dtype), (0, 6))
    off = 0
    for k, (o, i) in enumerate(EXPOSURE_SHAPES):
        h = h @ params[off:off + o * i].view(o, i).t()
        off += o * i
        if k < 3:
            h = torch.relu(h)
    return h[:, :12]


def affine_color_transform(params: Tensor, rgb: Tensor, hist: Tensor, batch_size: int) -> Tensor:
    """nerfh_nff.py:605-626: hist.long() -> exposure network -> rgb' = sigmoid(K rgb + b) per image."""
    a = exposure_mlp(params, hist.long())
    kernel, bias = a[:, :9].reshape(-1, 3, 3), a[:, 9:].reshape(-1, 3, 1)
    x = rgb.reshape(batch_size, -1, 3)
    x = torch.bmm(kernel, x.transpose(1, 2)) + bias
    return torch.sigmoid(x.transpose(1, 2).reshape(-1, 3))


FUSION_MEAN, FUSION_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]      # nerfh_nff.py:359-360


def fusion_net(sd: Dict[str, Tensor], rgb: Tensor, feat: Tensor, H: int, W: int, B: int, residual: bool = False, conv_pos=None,
               audit=None) -> Tensor:
    """run_fusion_net (:578-603) + FusionNet.forward (:395-418), BatchNorm2d in train mode (batch statistics, eps 1e-5).
    `sd`: the fusion_net state dict (net.0/2/4/6 conv weight+bias, net.7 BatchNorm weight+bias), any dtype.
    `conv_pos` (tests only): three boolean tensors, the ReLU branch pattern (output > 0) a HIP forward pass took in the three
    hidden layers -- the oracle is then evaluated ON that pattern, where the function is smooth (tests/branch.py's rule, extended to
    the loop's CNN); `audit`: dict that receives the number of units whose own sign differs and how far from zero they sit."""
    dt = rgb.dtype
    x = torch.cat([rgb.reshape(B, H, W, 3).permute(0, 3, 1, 2), feat.reshape(B, H, W, -1).permute(0, 3, 1, 2)], 1)
    mean, std = torch.tensor(FUSION_MEAN, dtype=dt), torch.tensor(FUSION_STD, dtype=dt)
    x = torch.cat([(x[:, :3] - mean[:, None, None]) / std[:, None, None], x[:, 3:]], 1)
    h = x
    for li, (k, pad) in enumerate(((0, 1), (2, 1), (4, 1), (6, 2))):
        h = F.conv2d(h, sd[f"net.{k}.weight"].to(dt), sd[f"net.{k}.bias"].to(dt), stride=1, padding=pad)
        if k != 6:
            if conv_pos is None:
                h = torch.relu(h)
            else:
                pos = conv_pos[li].reshape(h.shape)
                if audit is not None:
                    with torch.no_grad():
                        flips = (h > 0) != pos
                        audit["flips"] = audit.get("flips", 0) + int(flips.sum())
                        audit["units"] = audit.get("units", 0) + h.numel()
                        if bool(flips.any()):
                            audit["worst"] = max(audit.get("worst", 0.), float(h[flips].abs().max() / h.abs().max()))
                h = h * pos.to(dt)
    if "net.7.weight" in sd:
        mu = h.mean(dim=(0, 2, 3), keepdim=True)
        var = ((h - mu) ** 2).mean(dim=(0, 2, 3), keepdim=True)                 # biased, as BatchNorm normalises with
        h = (h - mu) / torch.sqrt(var + 1e-5)
        h = h * sd["net.7.weight"].to(dt)[None, :, None, None] + sd["net.7.bias"].to(dt)[None, :, None, None]
    return x[:, 3:] + h if residual else h


def feature_loss(feature_rgb: Tensor, feature_target: Tensor, per_pixel: bool = False) -> Tensor:
    """DFM_pose_refine.py:211-233 with img_in=True: [C,H,W] each."""
    C = feature_rgb.shape[0]
    a, b = feature_rgb.reshape(C, -1), feature_target.reshape(C, -1)
    return 1 - F.cosine_similarity(a, b, dim=0 if per_pixel else 1, eps=1e-6).mean()


class Adam:
    """torch.optim.Adam (defaults) on a list of (tensor, lr): same update order as torch's single-tensor path."""

    def __init__(self, params: List[Tensor], lrs: List[float], betas=(0.9, 0.999), eps: float = 1e-8):
        self.params, self.lrs, self.b1, self.b2, self.eps = params, lrs, betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.step_count = 0

    def step(self, grads: List[Tensor]):
        self.step_count += 1
        c1, c2 = 1 - self.b1 ** self.step_count, 1 - self.b2 ** self.step_count
        for p, g, m, v, lr in zip(self.params, grads, self.m, self.v, self.lrs):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / (c2 ** 0.5)).add_(self.eps)
            p.addcdiv_(m, denom, value=-lr / c1)


def structure_scene(named: Dict[str, Tensor], gain: float, decay: float, sigma_gain: float) -> None:
    """A synthetic scene with spatial structure, as a deterministic in-place edit of seed-0 random weights (`named`: parameter
    name -> tensor, e.g. dict(module.named_parameters()) under no_grad, or make_field_params' dict).  Default-init weights give
    a field that is almost constant in space (every layer shrinks the signal), so rendered features barely depend on the pose and
    the refinement loop of round 2's fixture was driven by a noise-level translation gradient.  Here: hidden-layer weights x
    `gain` (signal-preserving), the positional columns of frequency band k of the two layers that read the xyz embedding x
    2^(-decay k) (a spectrum that falls off like a trained field's instead of white noise down to 2 pi / 512), density head x
    `sigma_gain`.  tools/make_golden_refine50.py applies it to the reference's modules, the tests to the product's and the
    oracle's: all three start from the same seed-0 draw (tests/test_pose.py checksums)."""
    for name, p in named.items():
        if name.startswith(("xyz_encoding_", "dir_encoding", "transient_encoding")) and name.endswith("weight"):
            p.mul_(gain)
    for lname in ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight"):
        w = named[lname]                                             # columns: x(3), then per band k: sin(3), cos(3)
        for k in range(10):
            w[:, 3 + 6 * k: 9 + 6 * k] *= float(2.0 ** (-decay * k))
    named["static_sigma.0.weight"].mul_(sigma_gain)


def svd_reg(pose: Tensor) -> Tensor:
    """dm/DFM_pose_refine.py:119-129: nearest rotation of the 3x3 block, R = U V^T (out of place)."""
    u, _, v = torch.svd(pose[:, :3])
    return torch.cat([u @ v.transpose(-2, -1), pose[:, 3:]], 1)


def image_descriptor(img: Tensor) -> Tensor:
    """What the stand-in regression network sees of the query image [1,3,H,W]: its 2x2 average-pooled colours, 12 numbers."""
    return F.adaptive_avg_pool2d(img, 2).reshape(-1)


def tiny_apr(weight: Tensor, bias: Tensor, desc: Tensor, use_svd_reg: bool = True) -> Tensor:
    """Stand-in for the absolute-pose-regression CNN of train_on_batch (DFM_APR_refine.py:91, out of the path's scope): one
    Linear(12, 12) on the image descriptor -> [3,4], then svd_reg as inference_pose_regression applies it (PoseNet branch,
    DFM_pose_refine.py:148-151).  What matters for the path: the pose is a function of trainable parameters."""
    pose = (weight @ desc + bias).reshape(3, 4)
    return svd_reg(pose) if use_svd_reg else pose


def psnr(a: Tensor, b: Tensor) -> Tensor:
    """models/nerfh.py img2mse / mse2psnr."""
    return -10. * torch.log(torch.mean((a - b) ** 2)) / torch.log(torch.tensor(10., dtype=a.dtype))


def ssim(x: Tensor, y: Tensor) -> Tensor:
    """utils/utils.py:15-49 (7x7 mean filters on reflection-padded images), mean over the map."""
    pool = lambda t: F.avg_pool2d(t, 7, 1)
    x, y = F.pad(x, (3, 3, 3, 3), mode="reflect"), F.pad(y, (3, 3, 3, 3), mode="reflect")
    mx, my = pool(x), pool(y)
    sx, sy, sxy = pool(x * x) - mx * mx, pool(y * y) - my * my, pool(x * y) - mx * my
    n = (2 * mx * my + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    d = (mx * mx + my * my + 0.01 ** 2) * (sx + sy + 0.03 ** 2)
    return torch.clamp(n / d, 0, 1).mean()


def pose_error(pose_gt, pose_pred):
    """compute_pose_error_SE3 (dm/pose_model.py:75-92, eval.py:34-51): (|t - t'|, angle of R' R^T in degrees).  The reference
    takes the angle as the norm of cv2.Rodrigues' rotation vector; the same angle here from atan2(|axis part|, trace part)."""
    import numpy as np
    a, b = np.asarray(pose_gt, dtype=np.float64), np.asarray(pose_pred, dtype=np.float64)
    R = b[:3, :3] @ a[:3, :3].T
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.linalg.norm(a[:3, 3] - b[:3, 3])), float(np.degrees(np.arctan2(s, 0.5 * (np.trace(R) - 1.0))))


class Problem:
    """Everything one image's refinement needs, cast to `dtype` once.  `upsample=(H, W)`: the train_on_batch variant --
    `target` is [C,H,W] at full resolution, the fused features are bicubically up-sampled to it and both cropped by 10 px."""

    def __init__(self, p_coarse, p_fine, fusion_sd, exposure_params, cfg: O.RenderCfg, hwf, tinyscale: int, near: float,
                 far: float, init_c2w: Tensor, target: Tensor, hist: Tensor, world: dict, dtype=torch.float64, upsample=None):
        H, W, focal = hwf
        self.h, self.w, self.f = int(H // tinyscale), int(W // tinyscale), float(focal) / tinyscale
        cast = lambda d: {k: v.to(dtype) for k, v in d.items()}
        self.p_coarse, self.p_fine, self.fusion_sd = cast(p_coarse), cast(p_fine), cast(fusion_sd)
        self.exposure_params, self.init_c2w = exposure_params.to(dtype), init_c2w.to(dtype)
        self.target, self.hist = target.to(dtype), hist.to(dtype)
        self.cfg, self.near, self.far, self.world, self.dtype = cfg, near, far, world, dtype
        self.upsample = upsample

    def loss_at_pose(self, pose: Tensor, want_rgb: bool = False, **pin):
        """From the 3x4 pose in APR/COLMAP coordinates on: DFM_optimization_NFF (:311-337) / train_on_batch (:97-131).
        `pin` (tests only): fine_act / z_fine / coarse_act of ref_cpu.render and conv_pos / conv_audit of fusion_net -- the oracle
        on a GIVEN ReLU branch pattern and GIVEN sample depths (those a HIP forward pass took: tests/branch.py)."""
        conv_pos, conv_audit = pin.pop("conv_pos", None), pin.pop("conv_audit", None)
        pose = fix_coord_supp(pose, self.world["pose_scale"], self.world["move_all_cam_vec"], self.world["pose_scale2"])
        rgb, _, _, extras = O.render(self.h, self.w, self.f, self.p_coarse, self.p_fine, self.cfg, c2w=pose, near=self.near,
                                     far=self.far, hist=self.hist, **pin)
        rgb = affine_color_transform(self.exposure_params, rgb, self.hist, 1)
        fused = fusion_net(self.fusion_sd, rgb, extras["feat_map"], self.h, self.w, 1, conv_pos=conv_pos, audit=conv_audit)
        target = self.target
        if self.upsample is not None:
            fused = F.interpolate(fused, size=self.upsample, mode="bicubic")[:, :, 10:-10, 10:-10]      # :114, :126
            target = target[:, 10:-10, 10:-10]                                                            # :125
        loss = feature_loss(fused[0], target)
        if want_rgb:                                                   # the verification step's image (:117-118, :128)
            img = rgb.reshape(1, self.h, self.w, 3).permute(0, 3, 1, 2)
            return loss, F.interpolate(img, size=self.upsample, mode="bicubic")[:, :, 10:-10, 10:-10]
        return loss

    def loss(self, r: Tensor, t: Tensor, **pin) -> Tensor:
        """DFM_optimization_NFF (:310-337)."""
        return self.loss_at_pose(learn_pose(r, t, self.init_c2w), **pin)

    def loss_and_grad(self, r, t, **pin):
        """(loss, d loss / d [r, t] as one 6-vector) at the given pose parameters."""
        r = torch.as_tensor(r, dtype=self.dtype).clone().requires_grad_()
        t = torch.as_tensor(t, dtype=self.dtype).clone().requires_grad_()
        loss = self.loss(r, t, **pin)
        gr, gt = torch.autograd.grad(loss, [r, t])
        return loss.detach(), torch.cat([gr, gt])


def refine(prob: Problem, lr_r: float, lr_t: float, iters: int):
    """`iters` iterations of DFM_optimization_NFF (:310-341) for one image.  Returns dict(losses [iters], poses [iters,3,4],
    r [iters,3], t [iters,3], grads [iters,6]) -- the pose after each optimizer step, the loss and gradient before it."""
    r = torch.zeros(3, dtype=prob.dtype)
    t = torch.zeros(3, dtype=prob.dtype)
    opt = Adam([r, t], [lr_r, lr_t])
    out = {"losses": [], "poses": [], "r": [], "t": [], "grads": []}
    for _ in range(iters):
        loss, g = prob.loss_and_grad(r, t)
        opt.step([g[:3], g[3:]])
        out["losses"].append(loss)
        out["poses"].append(learn_pose(r, t, prob.init_c2w).clone())
        out["r"].append(r.clone())
        out["t"].append(t.clone())
        out["grads"].append(g)
    return {k: torch.stack(v) for k, v in out.items()}


def refine_apr(prob: Problem, weight: Tensor, bias: Tensor, photo: Tensor, lr: float, iters: int, use_svd_reg: bool = True):
    """`iters` iterations of train_on_batch (DFM_APR_refine.py:84-156) for one query image with the stand-in regression network
    `tiny_apr(weight, bias)` and Adam(lr) over its parameters (:212, :223).  `photo` [1,3,H,W] is the query image.  Returns
    dict(losses [iters], grads [iters,12] = d loss / d (the network's 12 outputs, before svd_reg), poses [iters,3,4] = the
    network's pose after each step, psnr / ssim [iters] of the verification step, final [3,4] = the pose the loop reports:
    the refined network's (:236-237), or the initial one when PSNR or SSIM ended lower than they started (:242-250))."""
    dt = prob.dtype
    W, b = weight.to(dt).clone(), bias.to(dt).clone()
    desc, photo = image_descriptor(photo.to(dt)), photo.to(dt)
    first = tiny_apr(W, b, desc, use_svd_reg)
    opt = Adam([W, b], [lr, lr])
    out = {"losses": [], "grads": [], "poses": [], "psnr": [], "ssim": []}
    for _ in range(iters):
        Wg, bg = W.clone().requires_grad_(), b.clone().requires_grad_()
        raw = Wg @ desc + bg
        pose = svd_reg(raw.reshape(3, 4)) if use_svd_reg else raw.reshape(3, 4)
        loss, img = prob.loss_at_pose(pose, want_rgb=True)
        gW, gb, graw = torch.autograd.grad(loss, [Wg, bg, raw])
        opt.step([gW, gb])
        out["losses"].append(loss.detach())
        out["grads"].append(graw)
        out["poses"].append(tiny_apr(W, b, desc, use_svd_reg))
        out["psnr"].append(psnr(img.detach(), photo[:, :, 10:-10, 10:-10]))
        out["ssim"].append(ssim(img.detach(), photo[:, :, 10:-10, 10:-10]))
    out = {k: torch.stack(v) for k, v in out.items()}
    retreat = bool(out["psnr"][-1] < out["psnr"][0]) or bool(out["ssim"][-1] < out["ssim"][0])
    out["final"] = first if retreat else out["poses"][-1]
    out["retreat"] = torch.tensor(retreat)
    return out
