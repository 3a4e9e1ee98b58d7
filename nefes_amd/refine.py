"""Per-image pose refinement loop around the HIP render path.

Mirrors the inner loop of script/dm/DFM_pose_refine.py:290-348 (`DFM_optimization_NFF`, `--pose_only 3`) and of
script/dm/DFM_APR_refine.py:84-156 (`train_on_batch`):

    LearnPose(cam) -> fix_coord_supp -> render(H/ts, W/ts, focal/ts, c2w) -> affine_color_transform(hist)
        -> run_fusion_net -> [bicubic upsample + 10 px crop] -> 1 - mean cosine similarity -> backward -> Adam

The reference runs the 50 iterations of one image as 50 eager Python passes (~150 launches each).  `PoseRefiner`
captures ONE iteration -- forward, backward and the Adam update -- into a HIP graph on first use and replays it, so the
loop's cost is the kernels' and not the host's; per-image state (pose delta, Adam moments, target features, histogram)
lives in static device buffers that are reset in place.  The bicubic up-sampling runs on ops.BicubicUpsample (gather
backward; the library's atomic scatter backward was 40 % of an iteration).  No CPU fallback anywhere on the path.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .pose import LearnPose
from .render import render, render_poses


def fix_coord_supp(pose, world_setup_dict, device=None):
    """dm/direct_pose_model.py:210-232: APR/COLMAP translation -> NeRF scale, `t' = (t*sc + move)*sc2`.
    Out of place (the reference mutates a clone), constants as device scalars so the chain is graph-capturable."""
    sc, sc2 = float(world_setup_dict['pose_scale']), float(world_setup_dict['pose_scale2'])
    mv = [float(v) * sc2 for v in world_setup_dict['move_all_cam_vec']]
    t = pose[..., :3, 3] * (sc * sc2)
    t = torch.stack([t[..., 0] + mv[0], t[..., 1] + mv[1], t[..., 2] + mv[2]], -1)
    return torch.cat([pose[..., :3, :3], t[..., None]], -1)


def feature_loss(feature_rgb, feature_target, img_in=True, per_pixel=False):
    """dm/DFM_pose_refine.py:211-233: 1 - mean cosine similarity (per channel over pixels, or per pixel over channels)."""
    if img_in:
        C = feature_rgb.shape[0]
        feature_rgb, feature_target = feature_rgb.reshape(C, -1), feature_target.reshape(C, -1)
    cos = nn.CosineSimilarity(dim=0 if per_pixel else 1, eps=1e-6)
    return 1 - cos(feature_rgb, feature_target).mean()


HIP_SVD_REG = os.environ.get("NEFES_HIP_SVD_REG", "1") != "0"


def svd_reg(pose):
    """dm/DFM_pose_refine.py:119-129: the 3x3 block of a regressed [B,3,4] pose replaced by its nearest rotation U V^T.
    Out of place (the reference writes into its argument).  fp32 poses on the GPU take ops.svd_reg (one launch each way, float64 inside,
    the polar factor's own derivative -- torch.svd there is a solver call plus a backward that divides by sigma_i^2 - sigma_j^2, ~0 for
    a near-rotation, and nothing of it can be captured into the iteration's graph); HIP_SVD_REG = False keeps the torch expression."""
    if HIP_SVD_REG and pose.is_cuda and pose.dtype == torch.float32:
        return ops.svd_reg(pose)
    u, _, v = torch.svd(pose[..., :3, :3])
    return torch.cat([u @ v.transpose(-2, -1), pose[..., :3, 3:]], -1)


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    """models/nerfh.py: -10 log10(mse)."""
    return -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))


def ssim_map(x, y):
    """utils/utils.py:15-49 (`SSIM`): 7x7 mean filters on reflection-padded images, clamped to [0, 1]."""
    F = nn.functional
    x, y = F.pad(x, (3, 3, 3, 3), mode="reflect"), F.pad(y, (3, 3, 3, 3), mode="reflect")
    pool = lambda t: F.avg_pool2d(t, 7, 1)
    mx, my = pool(x), pool(y)
    sx, sy, sxy = pool(x ** 2) - mx ** 2, pool(y ** 2) - my ** 2, pool(x * y) - mx * my
    n = (2 * mx * my + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    d = (mx ** 2 + my ** 2 + 0.01 ** 2) * (sx + sy + 0.03 ** 2)
    return torch.clamp(n / d, 0, 1)


class FeatureLoss(nn.Module):
    """dm/DFM_pose_refine.py:236-255."""

    def __init__(self, img_in=True, per_pixel=False):
        super().__init__()
        self.img_in, self.per_pixel = img_in, per_pixel

    def forward(self, feature_rgb, feature_target):
        return feature_loss(feature_rgb, feature_target, self.img_in, self.per_pixel)


class PoseRefiner:
    """`refine(init_c2w, feature_target, hist, iters)` -> (refined 4x4 c2w, losses [iters]) for one query image, or for
    `images=B` of them side by side (see refine()).

    hwf: full-resolution (H, W, focal); the render runs at 1/tinyscale.  `upsample=True` is the APR variant (bicubic
    upsample of the fused features to (H, W) and a 10-pixel crop; `feature_target` is [C, H, W]); False is the
    DFM_pose_refine variant (`feature_target` is [C, H/ts, W/ts]).  `world_setup` = dict(pose_scale, pose_scale2,
    move_all_cam_vec) or None.  `graph=True` replays one captured HIP graph per iteration.

    `pose_model=` selects `train_on_batch` proper (DFM_APR_refine.py:84-156, `pose_only=2`): the pose is what a regression
    network predicts from the query image (`pose_model(img)` -> [1,12] or [1,3,4]; `svd_reg=True` as config_stairs_DFM.txt:22),
    the optimizer is Adam(lr=`learning_rate`) over a deep copy of that network per image (:209-212), and `refine_apr` returns what
    `DFM_post_processing` records for the image: the refined network's pose, or the initial one when the verification step
    finds PSNR or SSIM of the up-sampled render lower after the loop than before (:233-250).  The network itself is the
    caller's (a CNN outside this path); implies `upsample=True`."""

    FUSED_UPSAMPLED_LOSS = True      # False: bicubic up-sampling and cosine loss as separate kernels (the tests compare the two)
    # The loop's target is fixed for all iterations of an image: its share of the up-sampled loss (Uy^T target Ux, |target|^2) is
    # computed when the target is set and an iteration runs on [C,h,w] data only (ops.UpcosTarget; csrc/refine.hip upcos_gram_*).
    # False: the one-pass kernels that read the whole target every iteration (the tests compare the two).
    PREPARED_TARGET = os.environ.get("NEFES_PREPARED_TARGET", "1") != "0"
    # Where the factored feature head renders the fine pass (ops.RenderFineFH: frozen width-128 network at test time), the head is not
    # applied per ray at all: render() hands over its input (65 channels) and FusionNet's first convolution runs on weights composed with
    # the head's (FusionNet.forward_prepared_gmap) -- the only consumer of the rendered features in this loop is that linear layer.
    GMAP_CONV0 = os.environ.get("NEFES_GMAP_CONV0", "1") != "0"
    # Mode 2: svd_reg + fix_coord_supp behind the regression network as one launch each way (ops.regressed_pose).  False: ops.svd_reg
    # followed by fix_coord_supp's torch expression (the tests compare the two and tap the pose in between).
    FUSED_REGRESSED_POSE = True
    FUSED_VERIFICATION = True       # PSNR + SSIM of the verification step as ops.psnr_ssim (False: the torch expressions)

    def __init__(self, render_kwargs, args, hwf, near, far, tinyscale=4, lr_r=0.01, lr_t=0.1, lietorch=False,
                 upsample=False, per_pixel=False, world_setup=None, graph=True, device="cuda", adam_capturable=None,
                 fused_glue=True, images=1, pose_model=None, svd_reg=False, learning_rate=1e-5, bn_running_stats=True):
        self.kw, self.args = dict(render_kwargs), args
        # False: FusionNet's train-mode BatchNorm leaves its running statistics and batch counter alone in this refiner's iterations
        # (nothing in the loop reads them; the reference never puts the net in eval mode).  REQUIRED for refiners that share a
        # FusionNet and run at the same time (refine_concurrently): the update is a read-modify-write of the shared module's buffers.
        self.bn_running_stats = bool(bn_running_stats)
        self.use_graph, self.fused_glue = bool(graph), bool(fused_glue)        # (read by _apr_adam below)
        H, W, focal = hwf
        self.H, self.W = int(H), int(W)
        self.h, self.w, self.focal = int(H // tinyscale), int(W // tinyscale), float(focal) / tinyscale
        self.near, self.far = float(near), float(far)
        self.upsample, self.per_pixel, self.world_setup = upsample, per_pixel, world_setup
        self.coarse = self.kw["network_fn"]
        self.C = self.coarse.W_features
        self.dev = torch.device(device)
        # images = B > 1: B query images refined side by side in ONE launch sequence per iteration (render_poses, a batched
        # FusionNet with per-image normalisation, per-image losses summed).  Every image's loss depends on its own six numbers
        # only and Adam is element-wise, so each image walks the trajectory it would walk alone; what is shared is the launch
        # overhead and the GPU (an 80x60 frame is 19 sample tiles per CU).  The reference refines one image at a time.
        self.B = int(images)
        if self.B > 1 and (not fused_glue or lietorch or per_pixel):
            raise NotImplementedError("nefes_amd: batched refinement runs on the fused glue kernels (fused_glue=True, "
                                      "lietorch=False, per_pixel=False)")
        self.apr = None
        if pose_model is not None:
            if self.B > 1 or per_pixel:
                raise NotImplementedError("nefes_amd: the regression-network variant refines one image at a time")
            import copy
            self.upsample = upsample = True
            self.apr_base, self.svd_reg = pose_model, bool(svd_reg)
            self.apr = copy.deepcopy(pose_model).to(self.dev).train()           # DFM_post_processing :209 (`pp_model`)
            self.photo = torch.zeros(1, 3, self.H, self.W, device=self.dev)
            self.apr_opt = self._apr_adam(learning_rate)
            self._rgb = self._x_rgb = None
        self.model = LearnPose(self.B, True, True, init_c2w=torch.eye(4)[None].repeat(self.B, 1, 1), lietorch=lietorch).to(self.dev)
        # fused_glue: the pose chain, the crop and the feature loss as library kernels (ops.pose_compose, the windowed
        # ops.bicubic_upsample, ops.cosine_feature_loss) and one fused Adam launch -- ~80 launches per iteration instead of ~215.
        # False keeps the torch expressions (the tests compare the two).
        self.fused_glue = bool(fused_glue)
        # fused glue: Adam over (r, t) with (lr_r, lr_t) as ONE launch (ops.FusedAdam: the two parameters and their gradients are
        # views of flat buffers, the pose kernel's backward writes the gradients there); otherwise torch.optim.Adam, two groups
        if self.fused_glue and not lietorch:
            self.opt = ops.FusedAdam([(self.model.r, lr_r), (self.model.t, lr_t)])
        else:
            self.opt = torch.optim.Adam([{"params": [self.model.r], "lr": lr_r}, {"params": [self.model.t], "lr": lr_t}],
                                        capturable=bool(graph) if adam_capturable is None else bool(adam_capturable),
                                        fused=True if self.fused_glue else None)
        th, tw = (self.H - 20, self.W - 20) if upsample else (self.h, self.w)
        self.target = torch.zeros(self.C, th, tw, device=self.dev) if self.B == 1 else torch.zeros(self.B, self.C, th, tw, device=self.dev)
        self.hist = torch.zeros(self.B, 10, device=self.dev)
        self.loss = torch.zeros((), device=self.dev) if self.B == 1 else torch.zeros(self.B, device=self.dev)
        self._affine = None            # static [1,12] buffer with the image's colour transform when the exposure network is frozen
        self.use_graph, self.graph, self.apr_graph = bool(graph), None, None

    # one iteration on the static buffers -------------------------------------------------------------------------
    def _loss(self):
        """DFM_optimization_NFF (:310-337): pose -> render -> affine colour transform -> fusion CNN -> feature loss.
        Returns (scalar to differentiate, per-image losses [B] or the same scalar)."""
        B = self.B
        if self.apr is not None:                           # train_on_batch :91-97
            c2w = self.apr(self.photo).reshape(1, 3, 4)
            if self.fused_glue and self.FUSED_REGRESSED_POSE and HIP_SVD_REG and c2w.is_cuda and c2w.dtype == torch.float32:
                c2w = ops.regressed_pose(c2w, self.svd_reg, self.world_setup)          # svd_reg + fix_coord_supp: one launch each way
            else:
                if self.svd_reg:
                    c2w = svd_reg(c2w)
                if self.world_setup is not None:
                    c2w = fix_coord_supp(c2w, self.world_setup)
        elif self.fused_glue and not self.model.lietorch:
            ws = self.world_setup or {"pose_scale": 1.0, "pose_scale2": 1.0, "move_all_cam_vec": (0., 0., 0.)}
            into = (self.model.r.grad, self.model.t.grad) if isinstance(self.opt, ops.FusedAdam) else None
            c2w = ops.pose_compose(self.model.r, self.model.t, self.model.init_c2w, ws["pose_scale"], ws["move_all_cam_vec"],
                                   ws["pose_scale2"], grad_into=into)                     # [B,3,4]
        else:
            c2w = self.model(0)[None, :3, :4]
            if self.world_setup is not None:
                c2w = fix_coord_supp(c2w, self.world_setup)
        enc = bool(getattr(self.args, "encode_hist", False))
        fnet = self.coarse.fusion_net
        fnet.track_bn_stats = self.bn_running_stats        # (read at call time by FusionNet._bn: eager calls and graph capture alike)
        fused_input = self.fused_glue and (not enc or self._affine is not None) and fnet._use_hip(self.hist)
        kw = self.kw
        if fused_input and self.GMAP_CONV0 and not fnet.fusion_residule and kw.get("network_fine", None) is not None:
            kw = dict(kw, feat_as_gmap=True)               # honoured only where render() takes the factored head (extras["feat_is_gmap"])
        if B == 1:
            # (a view, not c2w[0]: indexing costs a zero fill and a copy in the backward)
            rgb, _, _, ex = render(self.h, self.w, self.focal, c2w=c2w.reshape(3, 4), near=self.near, far=self.far, img_idx=self.hist, **kw)
            feat = ex["feat_map"]
        else:
            rgb, _, _, ex = render_poses(self.h, self.w, self.focal, c2w, near=self.near, far=self.far, **kw)
            rgb, feat = rgb.reshape(-1, 3), ex["feat_map"].reshape(-1, ex["feat_map"].shape[-1])
        gmap = bool(ex.get("feat_is_gmap", False))
        if fused_input:
            # colour transform + normalisation + the NCHW concatenation as one launch each way (ops.fusion_input)
            x = ops.fusion_input(rgb, feat, self._affine if enc else None, B, self.h, self.w, fnet.mean, fnet.std)
            if self.apr is not None:
                self._x_rgb = x.detach()[:, :3]            # the verification step's image, still normalised (:117-118)
            if gmap:
                _, w_f, _, b_f = kw["network_fine"].packed_fh()
                fused = fnet.forward_prepared_gmap(x, w_f, b_f, per_image_norm=B > 1)
            else:
                fused = fnet.forward_prepared(x, per_image_norm=B > 1)
        else:
            if enc:
                if self._affine is not None:               # frozen exposure network: its 12 numbers were computed once per image
                    rgb = self.coarse.apply_affine(self._affine, rgb, B)
                else:
                    rgb = self.coarse.affine_color_transform(self.args, rgb, self.hist, B)
            if self.apr is not None:
                self._rgb, self._x_rgb = rgb.detach(), None
            _, _, fused = self.coarse.run_fusion_net(rgb, feat, self.h, self.w, B, per_image_norm=B > 1)
        if self.upsample and self.fused_glue and not self.per_pixel and self.FUSED_UPSAMPLED_LOSS:
            # up-sampling, crop and feature loss in one pass each way: the 34 MB up-sampled image is never written (ops.upsampled_cosine_loss)
            if self._uses_prepared_target():
                mean_loss, cos = ops.upsampled_cosine_loss_prepared(fused.reshape(B * self.C, self.h, self.w), self._sync_upcos(), return_cos=True)
            else:
                mean_loss, cos = ops.upsampled_cosine_loss(fused.reshape(B * self.C, self.h, self.w),
                                                           self.target.reshape(B * self.C, self.H - 20, self.W - 20), (self.H, self.W), crop=10,
                                                           return_cos=True)
            if B > 1:
                return mean_loss * float(B), (1.0 - cos.view(B, self.C).mean(1)).float()
            return mean_loss, mean_loss.detach()
        if self.upsample:
            if self.fused_glue:
                fused = ops.bicubic_upsample(fused, (self.H, self.W), crop=10)
            else:
                fused = ops.bicubic_upsample(fused, (self.H, self.W))[:, :, 10:-10, 10:-10]
        if B > 1:
            # sum over images of (1 - mean_c cos) = B x (1 - mean over all B*C channels): one launch pair for the whole batch
            mean_loss, cos = ops.cosine_feature_loss(fused.reshape(B * self.C, -1), self.target.reshape(B * self.C, -1), return_cos=True)
            return mean_loss * float(B), (1.0 - cos.view(B, self.C).mean(1)).float()
        # (a view where possible: fused[0] costs a 34 MB zero fill and a copy in the backward)
        fused = fused.view(fused.shape[1:]) if fused.is_contiguous() else fused[0]
        if self.fused_glue and not self.per_pixel:
            loss = ops.cosine_feature_loss(fused, self.target)
        else:
            loss = feature_loss(fused, self.target, per_pixel=self.per_pixel)
        return loss, loss.detach()

    def _sync_upcos(self):
        """The prepared form of self.target (ops.UpcosTarget), recomputed whenever the target buffer was written since (tensor version
        counter).  _reset / refine_apr call this after copying an image's target in, i.e. before a captured graph is replayed; under
        capture the buffers must already be current (a re-computation would be replayed with every iteration)."""
        key = (self.target.data_ptr(), self.target._version)
        if getattr(self, "_upcos", None) is None or self._upcos_shape != tuple(self.target.shape):
            self._upcos = ops.UpcosTarget(self.B * self.C, self.h, self.w, self.H, self.W, 10, self.dev)
            self._upcos_shape, self._upcos_key = tuple(self.target.shape), None
        if self._upcos_key != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("nefes_amd: the refinement target changed without PoseRefiner._sync_upcos() before graph capture")
            with torch.no_grad():
                self._upcos.update(self.target.reshape(self.B * self.C, self.H - 20, self.W - 20))
            self._upcos_key = key
        return self._upcos

    def _uses_prepared_target(self):
        """The Gram-form loss against the prepared target -- where the loop takes the fused up-sampled loss at all AND the geometry fits
        the Gram kernel's LDS (ops.UpcosTarget.fits, known once the buffers exist); otherwise the one-pass kernels, which read
        self.target itself on every iteration."""
        if not (self.upsample and self.fused_glue and not self.per_pixel and self.FUSED_UPSAMPLED_LOSS and self.PREPARED_TARGET):
            return False
        if getattr(self, "_upcos", None) is None or self._upcos_shape != tuple(self.target.shape):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("nefes_amd: the refinement target changed without PoseRefiner._sync_upcos() before graph capture")
            self._upcos = ops.UpcosTarget(self.B * self.C, self.h, self.w, self.H, self.W, 10, self.dev)
            self._upcos_shape, self._upcos_key = tuple(self.target.shape), None
        return bool(self._upcos.fits)

    def loss_and_grad(self):
        """Loss at the current (r, t) and its gradient, written into the parameters' static .grad buffers (no optimizer step).
        The gradients are copied over the previous ones: nothing has to be zeroed between iterations."""
        loss, per_image = self._loss_into_buffer()
        if getattr(self, "_one", None) is None or self._one.shape != loss.shape:
            self._one = torch.ones_like(loss)              # the root gradient, once: autograd otherwise fills a new one per iteration
        gr, gt = torch.autograd.grad(loss, [self.model.r, self.model.t], grad_outputs=self._one)
        for p, g in ((self.model.r, gr), (self.model.t, gt)):
            if p.grad is None:
                p.grad = torch.empty_like(p)
            if g.data_ptr() != p.grad.data_ptr():          # (the fused pose kernel writes the parameters' .grad itself)
                p.grad.copy_(g)
        if per_image.data_ptr() != self.loss.data_ptr():   # (the fused loss kernels wrote it there themselves: _loss_into_buffer)
            self.loss.copy_(per_image)
        return self.loss

    def _loss_into_buffer(self):
        """self._loss() with the library's loss kernel writing its value straight into the static buffer self.loss (one image, fused glue):
        no 4-byte copy launch per iteration.  Everything else (images=B, the torch expressions) returns its own tensor and is copied."""
        ops.LOSS_OUT = self.loss if (self.B == 1 and self.fused_glue) else None
        try:
            return self._loss()
        finally:
            ops.LOSS_OUT = None

    def _iteration(self):
        self.loss_and_grad()
        self.opt.step()

    def _reset(self, init_c2w, feature_target, hist):
        with torch.no_grad():
            self.model.r.zero_()
            self.model.t.zero_()
            self.model.init_c2w.copy_(init_c2w.reshape(self.B, 4, 4))
            self.target.copy_(feature_target.reshape(self.target.shape))
            if self._uses_prepared_target():
                self._sync_upcos()
            self.hist.copy_(hist.reshape(self.B, 10))
            expo = getattr(self.coarse, "exposure_embedding", None)
            if (self.fused_glue and getattr(self.args, "encode_hist", False) and expo is not None
                    and not any(p.requires_grad for p in expo.parameters())):
                a = self.coarse.exposure_coefficients(self.hist)
                if self._affine is None:
                    self._affine = a.clone()
                else:
                    self._affine.copy_(a)                  # in place: a captured graph keeps reading the same buffer
            if isinstance(self.opt, ops.FusedAdam):
                self.opt.zero_state()
            else:
                for st in self.opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
                for p in (self.model.r, self.model.t):
                    if p.grad is not None:
                        p.grad.zero_()

    def _capture(self):
        """Warm up on a side stream (allocator, MIOpen plans, Adam state), then capture one iteration."""
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._iteration()
        self.graph = g

    def replay(self, apr=False):
        """One iteration as the captured graph -- after a HOST check that what the graph reads in place is current: the prepared form of
        self.target (ops.UpcosTarget) is refreshed when the target buffer was written since (its tensor version moved).  Code that
        writes refiner.target itself and replays refiner.graph directly optimises against the OLD target without any error (the
        one-pass kernels read self.target on every iteration; the prepared form cannot); this is the call to use instead."""
        if self._uses_prepared_target():
            self._sync_upcos()
        (self.apr_graph if apr else self.graph).replay()

    def refine(self, init_c2w, feature_target, hist, iters=50):
        """One image (images=1): (refined 4x4 c2w, losses [iters]).  A batch (images=B): init_c2w [B,4,4], feature_target
        [B,C,h,w], hist [B,10] -> (refined poses [B,4,4], losses [iters,B]), each image as if refined alone."""
        init_c2w, feature_target, hist = init_c2w.to(self.dev), feature_target.to(self.dev), hist.to(self.dev)
        self._reset(init_c2w, feature_target, hist)
        if self.use_graph and self.graph is None:
            self._capture()
            self._reset(init_c2w, feature_target, hist)
        losses = torch.empty((iters,) + tuple(self.loss.shape), device=self.dev)
        for i in range(iters):
            if self.use_graph:
                self.replay()
            else:
                self._iteration()
            losses[i] = self.loss
        with torch.no_grad():
            if self.B == 1:
                return self.model(0).detach().clone(), losses
            return self.model(torch.arange(self.B, device=self.dev)).detach().clone(), losses

    refine_batch = refine

    # ---- train_on_batch / DFM_post_processing (pose_only=2) ------------------------------------------------------------
    def apr_loss_and_grad(self):
        """Loss at the working network's current parameters; gradients into their .grad (no optimizer step)."""
        loss, _ = self._loss_into_buffer()
        params = [p for p in self.apr.parameters() if p.requires_grad]
        # loss.backward() in the reference (train_on_batch, DFM_pose_refine.py:318) tolerates parameters the loss never touches --
        # DFNet's adaptation_layers with return_feature=False (feature/dfnet.py:142): their .grad stays None and Adam skips them
        if getattr(self, "_one", None) is None or self._one.shape != loss.shape:
            self._one = torch.ones_like(loss)              # the root gradient, once (as loss_and_grad does)
        for p, g in zip(params, torch.autograd.grad(loss, params, grad_outputs=self._one, allow_unused=True)):
            p.grad = g
        if loss.data_ptr() != self.loss.data_ptr():
            self.loss.copy_(loss.detach())
        return self.loss

    def _verification(self, as_tensors=False):
        """PSNR and mean SSIM of the up-sampled, cropped render against the cropped query image (:117-128, :146-150)."""
        if self._x_rgb is not None:                        # de-normalise FusionNet's colour channels
            fnet = self.coarse.fusion_net
            mean, std = fnet._mean_std(self._x_rgb)
            img = self._x_rgb * std[:, None, None] + mean[:, None, None]
        else:
            img = self._rgb.reshape(1, self.h, self.w, 3).permute(0, 3, 1, 2)
        img = nn.functional.interpolate(img, size=(self.H, self.W), mode="bicubic")[:, :, 10:-10, 10:-10]
        gt = self.photo[:, :, 10:-10, 10:-10]
        if self.fused_glue and self.FUSED_VERIFICATION and img.is_cuda and img.dtype == torch.float32 and img.shape[0] == 1:
            ps, ss = ops.psnr_ssim(img[0], gt[0])          # two launches for PSNR + SSIM (torch: ~25); the crops stay views
        else:
            ps, ss = mse2psnr(img2mse(img, gt)).reshape(()), ssim_map(img, gt).mean().reshape(())
        if as_tensors:                                     # (no host read: the callers convert after their last replay)
            return ps, ss
        return float(ps), float(ss)

    def predicted_pose(self, net=None):
        """inference_pose_regression (DFM_pose_refine.py:131-160) of the working (or given) network on the query image: [3,4]."""
        with torch.no_grad():
            pose = (self.apr if net is None else net)(self.photo).reshape(1, 3, 4)
            return (svd_reg(pose) if self.svd_reg else pose)[0]

    def _apr_image_state(self, photo, feature_target, hist):
        """Per-image state of train_on_batch, written IN PLACE (a captured iteration keeps reading the same buffers): the query image, its
        cropped features (:125), a fresh copy of the regression network (:209), the histogram and the colour transform's twelve numbers."""
        dev = self.dev
        with torch.no_grad():
            self.photo.copy_(photo.to(dev).reshape(self.photo.shape))
            self.target.copy_(feature_target.to(dev).reshape(self.C, self.H, self.W)[:, 10:-10, 10:-10])
            if self._uses_prepared_target():
                self._sync_upcos()
            for pw, pb in zip(self.apr.parameters(), self.apr_base.parameters()):
                pw.copy_(pb.to(dev))                                       # a fresh copy of the network per image (:209)
            for bw, bb in zip(self.apr.buffers(), self.apr_base.buffers()):
                bw.copy_(bb.to(dev))                                       # deepcopy resets BatchNorm statistics too
            self.hist.copy_(hist.to(dev).reshape(1, 10))
            expo = getattr(self.coarse, "exposure_embedding", None)
            if (self.fused_glue and getattr(self.args, "encode_hist", False) and expo is not None
                    and not any(p.requires_grad for p in expo.parameters())):
                a = self.coarse.exposure_coefficients(self.hist)
                if self._affine is None:
                    self._affine = a.clone()
                else:
                    self._affine.copy_(a)

    def _apr_iteration(self):
        self.apr_loss_and_grad()
        self.apr_opt.step()

    def _apr_adam(self, lr):
        """torch.optim.Adam over the regression network's parameters (DFM_APR_refine.py:212).  On the GPU its multi-tensor form
        (`fused=True`: ONE launch for all parameters + one for the step counters; same update, same rounding order per element) instead
        of the default for-each form's ten element-wise launches per step -- the network is the caller's, so is the number of its
        parameter tensors (DFNet: ~100)."""
        params = list(self.apr.parameters())
        on_gpu = all(p.is_cuda for p in params)
        return torch.optim.Adam(params, lr=lr, capturable=bool(self.use_graph) and on_gpu, fused=True if (on_gpu and self.fused_glue) else None)

    def _apr_fresh_adam(self):
        """torch.optim.Adam(pp_model.parameters(), lr) per image (:212).  Under a graph the optimizer object is the captured one and its
        state is zeroed in place instead -- the same thing to Adam (step 0, zero moments)."""
        if self.apr_graph is None:
            self.apr_opt = self._apr_adam(self.apr_opt.param_groups[0]["lr"])
            return
        with torch.no_grad():
            for st in self.apr_opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()

    def _capture_apr(self):
        """One iteration of train_on_batch -- regression network, svd_reg, render, FusionNet, loss, backward to the network's weights,
        Adam -- as ONE HIP graph.  Possible since round 5: svd_reg is a kernel (torch.svd calls a solver and checks its status on the
        host), the target's share of the loss is prepared outside.  Warm-up on a side stream as _capture does; the caller re-writes the
        image's state afterwards (the warm-up steps moved the network)."""
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._apr_iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        for p in self.apr.parameters():
            p.grad = None                                   # the captured backward allocates them in the graph's pool and keeps them there
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._apr_iteration()
        self.apr_graph = g
        self._captured_apr_opt = self.apr_opt               # (the graph updates THIS optimizer's state tensors)

    def refine_apr(self, photo, feature_target, hist, iters=50, verification=True):
        """One query image, `pose_only=2`: `photo` [1,3,H,W], `feature_target` [C,H,W] (the query image's features at full
        resolution; cropped here as :125 does).  Returns (pose [3,4] in the regression network's coordinates, losses [iters],
        info = dict(psnr=(first, last), ssim=(first, last), retreat=bool)).  PoseRefiner(graph=True): the iteration is captured on
        first use and replayed (the regression network must be capturable: no host reads, static shapes)."""
        if self.apr is None:
            raise RuntimeError("nefes_amd: PoseRefiner was built without pose_model=")
        dev = self.dev
        self._apr_image_state(photo, feature_target, hist)
        self._apr_fresh_adam()
        if self.use_graph and self.apr_graph is None:
            self._capture_apr()
            self._apr_image_state(photo, feature_target, hist)
            self._apr_fresh_adam()
        first = self.predicted_pose()
        losses = torch.empty(iters, device=dev)
        checks = []
        for i in range(iters):
            if self.apr_graph is not None:
                self.replay(apr=True)
            else:
                self._apr_iteration()
            losses[i] = self.loss
            if verification and (i == 0 or i == iters - 1):
                checks.append(self._verification(as_tensors=True))        # (read after the loop: no host sync behind iteration 0)
        pose = self.predicted_pose()
        info = {"retreat": False}
        if verification and len(checks) == 2:
            (p0, s0), (p1, s1) = [(float(a), float(b)) for a, b in checks]
            info = {"psnr": (p0, p1), "ssim": (s0, s1), "retreat": bool(p1 < p0) or bool(s1 < s0)}       # :242-250
            if info["retreat"]:
                pose = first
        return pose.detach().clone(), losses, info


def _apr_kernels_bit_stable(r, photo, feature_target, hist, trials=6):
    """EMPIRICAL guard for refine_apr_concurrently.  The regression network, its backward, torch's Adam and the verification step are the
    CALLER'S / torch's kernels; next to another stream's field kernels such code is exposed to the packed-fp32 finding of DESIGN.md 4.7
    (hipcc vectorises torch's kernels the way it vectorised ours) -- whether a given network's kernels contain the affected forms cannot be
    read off its Python.  So it is tried: two eager iterations + the verification step from the image's initial state, alone on the
    device and `trials` times with this library's fine-field forward running on a second stream, every result compared bit for bit
    (loss, every parameter of the working network, PSNR, SSIM).  The affected instructions go wrong in 0.05-0.35 % of their executions on
    a quarter of the lanes: a kernel that carries one does not survive six trials of this.  Cached on the network object."""
    key = "_nefes_bit_stable_next_to_field_kernels"
    if getattr(r.apr_base, key, None) is not None:
        return getattr(r.apr_base, key)
    from . import lib as L
    dev = r.dev
    fine = r.kw["network_fine"]
    pk = fine.packed()
    g = torch.Generator().manual_seed(5)
    N, S = 4800, 128
    ro = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
    rd = nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
    z = (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(dev)
    side = torch.cuda.Stream(device=dev)

    def run(storm):
        r._apr_image_state(photo, feature_target, hist)
        r._apr_fresh_adam()
        torch.cuda.synchronize(dev)
        keep = []
        if storm:
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(8):                         # ~5 ms of field kernels: longer than the sequence under test
                    keep.append(ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL))
        r._apr_iteration()
        r._apr_iteration()
        ps, ss = r._verification(as_tensors=True)
        out = [r.loss.detach().clone(), ps.detach().clone(), ss.detach().clone()] + [p.detach().clone() for p in r.apr.parameters()]
        torch.cuda.synchronize(dev)
        return out

    was_graph = r.apr_graph
    views = (r._rgb, r._x_rgb)                             # (a captured iteration's views of ITS static buffers: eager calls re-point them)
    r.apr_graph = None                                     # (eager iterations; _apr_fresh_adam then builds a fresh optimizer per run)
    try:
        solo = run(False)
        ok = all(all(torch.equal(a, b) for a, b in zip(solo, run(True))) for _ in range(trials))
    finally:
        r.apr_graph = was_graph
        if was_graph is not None:                          # the captured optimizer object and buffer views must be the live ones again
            r.apr_opt = r._captured_apr_opt
            r._rgb, r._x_rgb = views
    setattr(r.apr_base, key, bool(ok))
    return bool(ok)


def refine_apr_concurrently(refiners, jobs, iters=50, verification=True, verify_kernels=True):
    """refine_apr for K query images at the same time (the shipped default mode, `pose_only = 2`): one PoseRefiner(pose_model=...) and
    one HIP stream per image, iterations replayed round-robin like refine_concurrently.  jobs = [(photo, feature_target, hist), ...] ->
    [(pose [3,4], losses [iters], info), ...] in job order, each identical to what refine_apr returns for that image alone.
    The regression network's kernels are the caller's: `verify_kernels` first TRIES them next to this library's field kernels
    (_apr_kernels_bit_stable) and raises if a result moves -- run such a network's images one after the other (refine_apr)."""
    if not refiners or not jobs:
        raise ValueError("nefes_amd: refine_apr_concurrently needs at least one PoseRefiner and one job")
    if len({id(r) for r in refiners}) != len(refiners):
        raise ValueError("nefes_amd: refine_apr_concurrently needs DISTINCT PoseRefiner objects")
    if any(r.apr is None for r in refiners):
        raise RuntimeError("nefes_amd: refine_apr_concurrently needs refiners built with pose_model=")
    if len(jobs) > len(refiners):
        out = []
        for k in range(0, len(jobs), len(refiners)):
            part = jobs[k:k + len(refiners)]
            out += refine_apr_concurrently(refiners[:len(part)], part, iters, verification, verify_kernels)
        return out
    refiners = refiners[:len(jobs)]
    dev = refiners[0].dev
    _check_concurrent(refiners, apr=True)
    if verify_kernels:
        for r, job in zip(refiners, jobs):
            if not _apr_kernels_bit_stable(r, *job):
                raise RuntimeError("nefes_amd: this regression network's kernels (or torch's Adam / the verification step) do not give "
                                   "bit-identical results next to another stream's field kernels (DESIGN.md 4.7): refine its images "
                                   "one after the other with PoseRefiner.refine_apr")
    for r, job in zip(refiners, jobs):                      # per-image state, capture (one refiner at a time, on the caller's stream)
        r._apr_image_state(*job)
        r._apr_fresh_adam()
        if r.use_graph and r.apr_graph is None:
            r._capture_apr()
    cur = torch.cuda.current_stream(dev)
    for r in refiners:
        if getattr(r, "_own_stream", None) is None:
            r._own_stream = torch.cuda.Stream(device=dev)
        r._own_stream.wait_stream(cur)
    firsts, losses, checks = [], [torch.empty(iters, device=dev) for _ in refiners], [[] for _ in refiners]
    for r, job in zip(refiners, jobs):
        with torch.cuda.stream(r._own_stream):
            r._apr_image_state(*job)
            r._apr_fresh_adam()
            firsts.append(r.predicted_pose())
    for i in range(iters):
        for k, r in enumerate(refiners):
            with torch.cuda.stream(r._own_stream):
                if r.apr_graph is not None:
                    r.replay(apr=True)
                else:
                    r._apr_iteration()
                losses[k][i] = r.loss
                if verification and (i == 0 or i == iters - 1):
                    checks[k].append(r._verification(as_tensors=True))
    poses = []
    for r in refiners:
        with torch.cuda.stream(r._own_stream):
            poses.append(r.predicted_pose().detach().clone())
    for r in refiners:
        cur.wait_stream(r._own_stream)
    torch.cuda.synchronize(dev)
    outs = []
    for k in range(len(refiners)):
        info, pose = {"retreat": False}, poses[k]
        if verification and len(checks[k]) == 2:
            (p0, s0), (p1, s1) = [(float(a), float(b)) for a, b in checks[k]]
            info = {"psnr": (p0, p1), "ssim": (s0, s1), "retreat": bool(p1 < p0) or bool(s1 < s0)}
            if info["retreat"]:
                pose = firsts[k]
        outs.append((pose.detach().clone(), losses[k], info))
    return outs


def _check_concurrent(refiners, apr=False):
    """What refine_concurrently's guarantees rest on, checked instead of assumed.
    (1) No shared mutable state between streams: refiners that share a FusionNet must not update its BatchNorm's running statistics
        (a read-modify-write of the module's buffers from several streams: lost counter increments, non-deterministic statistics).
    (2) Every kernel of an iteration is this library's: the bit-identity of concurrent and solo trajectories holds for kernels built
        without the packed-fp32 op_sel:[0,1] forms; MIOpen / ATen kernels next to another stream's MFMA kernels are
        exposed to it.  A refiner on the torch glue, torch convolutions / BatchNorm or torch.optim.Adam is refused here."""
    shared = {}
    for r in refiners:
        fnet = r.coarse.fusion_net
        bn_tracks = (not fnet.no_BN) and r.B == 1 and fnet.net[-1].training and fnet.net[-1].track_running_stats
        other = shared.setdefault(id(fnet), r)
        if other is not r and bn_tracks and (r.bn_running_stats or other.bn_running_stats):
            raise ValueError("nefes_amd: refine_concurrently: these refiners share one FusionNet whose BatchNorm would update its running "
                             "statistics from several streams at once; construct them with PoseRefiner(..., bn_running_stats=False)")
        # (apr: the regression network and its optimizer are torch's by construction; refine_apr_concurrently tries them instead)
        in_house = (r.fused_glue and (apr or isinstance(r.opt, ops.FusedAdam)) and fnet.HIP_CONVS and (fnet.no_BN or fnet.HIP_BATCHNORM)
                    and fnet._use_hip(r.hist))
        if not in_house:
            raise ValueError("nefes_amd: refine_concurrently needs every refiner on the library's own kernels (fused_glue=True, "
                             "lietorch=False, FusionNet.HIP_CONVS / HIP_BATCHNORM, frozen FusionNet on the GPU): torch's kernels next to "
                             "another stream's field kernels are exposed to the packed-fp32 finding of DESIGN.md 4.7; run such "
                             "refiners one after the other (PoseRefiner.refine)")


def refine_concurrently(refiners, jobs, iters=50):
    """K query images refined AT THE SAME TIME: one PoseRefiner (its own static buffers and captured graph; the frozen networks can be
    shared) and one HIP stream per image, replays issued round-robin.  An iteration is ~1.45 ms of field kernels, which hold the whole
    chip at its power limit, and ~0.3 ms of FusionNet convolutions, loss and pose kernels that are bound by launch and L2 latency on a
    fraction of the CUs: alone on the device those 0.3 ms are idle silicon, next to another image's field kernels they are nearly
    free (two images: -8 % per image; tools/two_streams.py).  Every image walks exactly the trajectory it walks alone -- bit for bit
    (tests/test_gpu_streams.py; that test is also what found the packed-fp32 op_sel instruction of DESIGN.md 4.7).
    jobs = [(init_c2w, feature_target, hist), ...] as PoseRefiner.refine takes them -> [(refined c2w, losses [iters]), ...] in job order.
    More jobs than refiners: rounds of len(refiners) images (the last one smaller).
    Refiners that share a FusionNet are constructed with bn_running_stats=False, and every refiner runs on the library's own kernels
    (_check_concurrent refuses anything else: see there)."""
    if not refiners or not jobs:
        raise ValueError("nefes_amd: refine_concurrently needs at least one PoseRefiner and one job")
    if len({id(r) for r in refiners}) != len(refiners):
        raise ValueError("nefes_amd: refine_concurrently needs DISTINCT PoseRefiner objects (each owns the static buffers of its image)")
    if len(jobs) > len(refiners):
        out = []
        for k in range(0, len(jobs), len(refiners)):
            part = jobs[k:k + len(refiners)]
            out += refine_concurrently(refiners[:len(part)], part, iters)
        return out
    refiners = refiners[:len(jobs)]
    dev = refiners[0].dev
    _check_concurrent(refiners)
    jobs = [tuple(t.to(dev) for t in job) for job in jobs]
    for r, job in zip(refiners, jobs):
        if r.use_graph and r.graph is None:
            r.refine(*job, iters=0)                        # reset, warm-up, capture (on the caller's stream, one refiner at a time)
    cur = torch.cuda.current_stream(dev)
    for r in refiners:
        if getattr(r, "_own_stream", None) is None:
            r._own_stream = torch.cuda.Stream(device=dev)
        r._own_stream.wait_stream(cur)
    losses = [torch.empty((iters,) + tuple(r.loss.shape), device=dev) for r in refiners]
    for r, job in zip(refiners, jobs):
        with torch.cuda.stream(r._own_stream):
            r._reset(*job)
    for i in range(iters):
        for r, l in zip(refiners, losses):
            with torch.cuda.stream(r._own_stream):
                if r.use_graph:
                    r.replay()
                else:
                    r._iteration()
                l[i] = r.loss
    outs = []
    for r, l in zip(refiners, losses):
        with torch.cuda.stream(r._own_stream), torch.no_grad():
            pose = r.model(0) if r.B == 1 else r.model(torch.arange(r.B, device=dev))
            outs.append((pose.detach().clone(), l))
    for r in refiners:
        cur.wait_stream(r._own_stream)
    return outs
