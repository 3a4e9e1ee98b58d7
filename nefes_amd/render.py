"""render()/render_rays()/batchify_rays()/sample_pdf() with the reference's signatures
(script/models/rendering.py:23-243), executed by the HIP kernels of libnefes_hip.so.

Kernel sequence of one render() at test time (perturb=0, test_time=True):
    RayGen (get_rays + viewdirs)                 -> rays_o, rays_d, viewdirs          [grad -> c2w]
    coarse_depths                                -> z_coarse [N,Nc]
    field_fwd SIGMA (coarse net, no grad)        -> sigma [N,1,Nc]
    composite_fwd variant D                      -> weights [N,Nc]
    sample_pdf_merge                             -> z_fine [N,Nc+Ni]
    FieldFromRays FULL (fine net, masks saved)   -> raw_t [N,R,S]                      [grad -> rays]
    Composite variant A/B                        -> rgb, feat, disp, acc               [grad -> raw_t]
The ray batch is processed in one launch per kernel (no Python chunk loop over 32 768-ray chunks as in
rendering.py:182-195): the batch is split only when its intermediates (raw_t, its gradient, the ReLU masks, train-mode
activations) would not fit the device memory -- see `rays_per_launch` -- or at MAX_RAYS_PER_LAUNCH.  `netchunk` is
accepted and ignored (the fused kernel tiles internally).
"""
import types

import torch

from . import lib as L
from . import ops

_DEV = "cuda"
MAX_RAYS_PER_LAUNCH = 1 << 22
MEMORY_FRACTION = 0.6          # share of the free device memory one launch sequence may plan for


_STEP = {}
_PER_RAY = {}


def rays_per_launch(cfg, network_fn, network_fine, device, train=False):
    """Largest ray count whose per-launch intermediates fit: per ray and fine sample 2 x R floats (raw_t and its gradient),
    the 1-bit ReLU masks, d pts / d viewdirs, and in train mode the saved pre-activations (layout.h row map); plus the
    depth / weight rows.  Bounded by MAX_RAYS_PER_LAUNCH.  Planned once per (network shape, sample counts, mode, device) against
    the device's TOTAL memory, so that where a batch is split -- and with it the order in which jitter / noise is drawn -- does
    not depend on what else happens to be allocated at the moment of a call (a plan that no longer fits fails in the allocator
    with torch's out-of-memory error)."""
    net = network_fine if (network_fine is not None and cfg.N_importance > 0) else network_fn
    S = cfg.N_samples + cfg.N_importance if not cfg.use_fine_only or cfg.N_importance == 0 else cfg.N_importance
    Wd, C = net.W, net.W_features
    dev = torch.device(device)
    key = (Wd, C, S, cfg.N_samples, bool(train), dev.type, dev.index)
    if key in _STEP:
        return _STEP[key]
    R = 3 + C + 6
    per_sample = 2 * R * 4 + (8 * (Wd // 64) + 4 * (Wd // 128)) * 4 * 2 + 24 + 16
    if train:
        # rows per sample of `acts` and `dacts` (csrc/layout.h nefes_train_row(..., NEFES_TB_END)): the head block is sized by the
        # head CLASS (nefes_head_ntr: one tile for 3 + C <= 32, five otherwise), not by C
        per_sample += 4 * (64 + 32 + 9 * Wd + 2 * Wd + 32 * (1 if 3 + C <= 32 else 5) + 64) * 2
    per_ray = per_sample * S + (cfg.N_samples * 4) * 4 + 256
    _PER_RAY[key] = per_ray
    try:
        total = torch.cuda.get_device_properties(dev).total_memory
    except Exception:
        return MAX_RAYS_PER_LAUNCH
    _STEP[key] = int(max(1024, min(MAX_RAYS_PER_LAUNCH, (total * MEMORY_FRACTION) // per_ray)))
    return _STEP[key]


def _clamp_to_free_memory(step, cfg, network_fn, network_fine, device):
    """Inference without jitter or noise draws nothing from the RNG, so where the batch is split changes no result: there the
    deterministic plan is additionally bounded by what is FREE right now (the feature extractor, APR copies, graph pools or
    another rank may share the device), instead of handing the allocator one launch it cannot serve."""
    net = network_fine if (network_fine is not None and cfg.N_importance > 0) else network_fn
    S = cfg.N_samples + cfg.N_importance if not cfg.use_fine_only or cfg.N_importance == 0 else cfg.N_importance
    dev = torch.device(device)
    per_ray = _PER_RAY.get((net.W, net.W_features, S, cfg.N_samples, False, dev.type, dev.index))
    if per_ray is None:
        return step
    try:
        free, _ = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # torch's own cache is reusable
    except Exception:
        return step
    return int(max(1024, min(step, (free * 0.8) // per_ray)))


def _cfg(kwargs):
    a = kwargs.get("args", None)
    g = lambda name, default: getattr(a, name, default) if a is not None else default
    return types.SimpleNamespace(
        N_samples=int(kwargs["N_samples"]), N_importance=int(kwargs.get("N_importance", 0)),
        perturb=float(kwargs.get("perturb", 0.)), lindisp=bool(kwargs.get("lindisp", False)),
        white_bkgd=bool(kwargs.get("white_bkgd", False)), raw_noise_std=float(kwargs.get("raw_noise_std", 0.)),
        test_time=bool(kwargs.get("test_time", False)), nerfh_nff=bool(g("nerfh_nff", True)),
        use_fine_only=bool(g("use_fine_only", False)), NeRFW=bool(g("NeRFW", True)),
        transient_at_test=bool(g("transient_at_test", False)),
        # BASELINE config 4: a hash-grid (ops.HashGrid) in front of the same MLP instead of the frequency embedding
        xyz_encoder=kwargs.get("xyz_encoder", None),
        # nefes_amd-private (the refinement loop sets it): where the factored feature head applies, return its per-ray INPUT
        # (sum_s w_s g_s, sum_s w_s) [N, W/2 + 1] as "feat_map" and say so with extras["feat_is_gmap"]; the caller folds the head into the
        # linear layer that consumes the features (FusionNet's first convolution)
        feat_as_gmap=bool(kwargs.get("feat_as_gmap", False)))


def _trainable(net):
    """True when a render would run the train-mode instances for `net`: autograd is recording AND one of the field's own weights
    asks for a gradient (the FusionNet / exposure sub-modules hang off the same nn.Module but are not part of the field)."""
    return (torch.is_grad_enabled() and net is not None
            and any(p.requires_grad for n, p in net.named_parameters()
                    if not n.startswith(("fusion_net", "exposure_embedding"))))


def _field(pk, mode, rays_o, rays_d, viewdirs, z, xyz_encoder):
    """raw_t [N,R,S] for samples z along the rays; differentiable w.r.t. rays_o, rays_d, viewdirs in FULL mode."""
    if xyz_encoder is None:
        return ops.FieldFromRays.apply(rays_o, rays_d, viewdirs, z, pk, mode)
    if mode != L.FIELD_STATIC and ops.hashgrid_fused_ok(pk, xyz_encoder) and z.shape[0] * z.shape[1] < (1 << 31) - 256:
        # BASELINE configs[3]: the field kernels gather the hash grid themselves (no [M, 32] encoding, no pts tensor)
        return ops.FieldFromRaysHashGrid.apply(rays_o, rays_d, viewdirs, z, pk, mode, xyz_encoder)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]                    # rendering.py:114,142 (torch glue)
    return ops.FieldFromEncoding.apply(xyz_encoder(pts), viewdirs, pk, mode)


def _render_core(rays_o, rays_d, viewdirs, near, far, network_fn, network_fine, cfg, bounds=None):
    """rendering.py:88-180 for n rays already on the GPU.  Returns the reference's `ret` dict.
    `bounds` ([n, >=2] rows starting with near, far) overrides the scalars, as the packed ray batch does (:90-93)."""
    N = rays_o.shape[0]
    dev = rays_o.device
    Nc, Ni = cfg.N_samples, cfg.N_importance
    # Trainable NeRF weights (run_nefes.py) go through the train-mode instances: forward with saved pre-activations and
    # the weight-gradient kernels of csrc/train.hip.  Frozen weights (refinement loop, DFM_APR_refine.py:192-193) use the
    # fused backward-to-rays path.
    trainable = _trainable

    def field(net, pk, mode, z_):
        if trainable(net) and mode != L.FIELD_SIGMA:
            if cfg.xyz_encoder is not None:
                raise NotImplementedError("nefes_amd: train mode is built for the frequency embedding only")
            from . import train as T
            return T.field_train(net, mode, rays_o, rays_d, viewdirs, z_)
        return _field(pk, mode, rays_o, rays_d, viewdirs, z_, cfg.xyz_encoder)

    pk_c = network_fn.packed()
    C = pk_c.feat_dim
    store_rgb = (Ni == 0)
    if (cfg.test_time and cfg.perturb == 0. and cfg.raw_noise_std == 0. and bounds is None
            and not cfg.use_fine_only and ops.fused_coarse_pass_ok(pk_c, Nc, Ni, N, cfg.xyz_encoder)):
        # The coarse pass at test time (:96-141, nerfh_nff.py:192-202) as TWO launches: every ray shares one row of depths, which is
        # computed once per (near, far, Nc) and never expanded; the sigma-only field kernel reads it; compositing variant D,
        # sample_pdf and the sort run per ray in one kernel, the coarse weights stay in registers.
        with torch.no_grad():
            z_row = ops.coarse_depth_row(Nc, near, far, cfg.lindisp, dev)
            raw_c = ops.field_sigma_row(pk_c, rays_o.detach(), rays_d.detach(), z_row, cfg.xyz_encoder)
            z_fine, z_samples = ops.coarse_sample(raw_c, z_row, Ni, want_samples=False)
        return _fine_pass(rays_o, rays_d, viewdirs, z_fine, z_samples, network_fine, cfg, C, field, None)
    t_rand = torch.rand(N, Nc, device=dev) if cfg.perturb > 0. else None            # :110 (RNG stays in torch)
    z = ops.coarse_depths(N, Nc, near, far, cfg.lindisp, t_rand, device=dev, bounds=bounds)
    if cfg.test_time:
        # coarse + test_time: sigma-only branch, nothing differentiable (nerfh_nff.py:192-202)
        with torch.no_grad():
            raw_c = _field(pk_c, L.FIELD_SIGMA, rays_o.detach(), rays_d.detach(), viewdirs.detach(), z, cfg.xyz_encoder)
            if cfg.raw_noise_std > 0.:
                raw_c = raw_c + _noise(raw_c.shape, dev) * cfg.raw_noise_std
            _, _, _, acc0, _, w0, _ = ops.composite_fwd(raw_c, z, C, L.COMP_SIGMA_ONLY)
        rgb0 = feat0 = disp0 = None
    else:
        raw_c = field(network_fn, pk_c, L.FIELD_STATIC, z)
        if cfg.raw_noise_std > 0.:
            raw_c = _with_density_noise(raw_c, C, cfg.raw_noise_std)
        rgb0, feat0, disp0, acc0, _, w0, _ = ops.Composite.apply(raw_c, z, C, 0, 0.1)
    if Ni == 0:
        if cfg.test_time:
            raise NotImplementedError("nefes_amd: N_importance == 0 at test time renders nothing in the reference "
                                      "either (sigma-only coarse branch, rendering.py:116-125)")
        ret = {"rgb_map": rgb0, "disp_map": disp0, "acc_map": acc0}
        if cfg.nerfh_nff:
            ret["feat_map"] = feat0
        return ret
    # hierarchical sampling (:132-141); z_samples are detached in the reference
    u = None if cfg.perturb == 0. else torch.rand(N, Ni, device=dev)
    z_fine, z_samples = ops.sample_pdf_merge(z, w0.detach(), Ni, u=u)
    return _fine_pass(rays_o, rays_d, viewdirs, z_fine, z_samples, network_fine, cfg, C, field, (rgb0, feat0, disp0, acc0))


def _noise(shape, device):
    """torch.randn_like(static_sigmas) of raw2outputs_NeRFH_NFF (nerfh_nff.py:67); a function of its own so that a test can put a
    known tensor in its place (the device's generator and the reference's CPU generator share no stream)."""
    return torch.randn(tuple(shape), device=device)


def _with_density_noise(raw, C, std):
    """raw [N, R, S] channel-major with the static density at channel 3 + C: density + randn * std (nerfh_nff.py:67-68)."""
    noise = torch.zeros_like(raw)
    noise[:, 3 + C] = _noise((raw.shape[0], raw.shape[2]), raw.device) * std
    return raw + noise


def _fine_pass(rays_o, rays_d, viewdirs, z_fine, z_samples, network_fine, cfg, C, field, coarse_maps):
    """rendering.py:142-180: the fine network at the merged depths, compositing, and the reference's `ret` dict.  coarse_maps =
    (rgb0, feat0, disp0, acc0) of a differentiable coarse pass, or None (test time)."""
    rgb0, feat0, disp0, acc0 = coarse_maps if coarse_maps is not None else (None, None, None, None)
    z_f = z_samples if cfg.use_fine_only else z_fine
    flags = 0
    if (cfg.test_time and cfg.NeRFW and cfg.nerfh_nff and cfg.xyz_encoder is None and cfg.raw_noise_std == 0.
            and z_f.shape[0] * z_f.shape[1] < (1 << 31) - 256 and network_fine.factored_head_ok()):
        # FACTORED HEAD (frozen width-128 network, test time).  The rgb+feature head is linear in g = relu(dir_encoding) and compositing is
        # linear in the head's outputs with weights that do not depend on them (nerfh_nff.py:119-125, :487-490):
        #     feat = sum_s w_s (W_f g_s + b_f) = W_f (sum_s w_s g_s) + b_f sum_s w_s.
        # The field kernels emit g (64 channels) + a channel of ones instead of the 128 feature channels, the compositor runs on 65
        # "features", and W_f is applied once per ray: 63 of 137 raw channels and 44 of the head's 60 MFMAs per 32 samples go.
        pk_fh, w_f, w_f_t, b_f = network_fine.packed_fh()
        flags |= L.COMP_TRANSIENT
        if not cfg.transient_at_test:
            flags |= L.COMP_STATIC_ONLY
        if cfg.white_bkgd:
            flags |= L.COMP_WHITE_BKGD
        rgb, feat, disp, acc = ops.RenderFineFH.apply(rays_o, rays_d, viewdirs, z_f, pk_fh, w_f, w_f_t, b_f, flags, float(network_fine.beta_min),
                                                      cfg.feat_as_gmap)
        ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "feat_map": feat}
        if cfg.feat_as_gmap:
            ret["feat_is_gmap"] = True
        return ret
    pk_f = network_fine.packed()
    mode = L.FIELD_FULL if cfg.NeRFW else L.FIELD_STATIC
    raw_f = field(network_fine, pk_f, mode, z_f)
    if cfg.raw_noise_std > 0. and not cfg.NeRFW:
        # the reference draws the noise in every composite WITHOUT the transient head (nerfh_nff.py:66-68) -- the coarse pass above and a
        # fine network with NeRFW off; with the transient head the line is not reached (:61-64)
        raw_f = _with_density_noise(raw_f, C, cfg.raw_noise_std)
    if cfg.NeRFW:
        flags |= L.COMP_TRANSIENT
        if cfg.test_time and not cfg.transient_at_test:
            flags |= L.COMP_STATIC_ONLY
    if cfg.white_bkgd:
        flags |= L.COMP_WHITE_BKGD
    rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw_f, z_f, C, flags, float(network_fine.beta_min))
    ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc}
    if cfg.nerfh_nff:
        ret["feat_map"] = feat
    if not cfg.test_time:                                                            # :160-173
        ret["rgb0"], ret["disp0"], ret["acc0"] = rgb0, disp0, acc0
        ret["z_std"] = torch.std(z_samples, dim=-1, unbiased=False)
        if cfg.NeRFW:
            ret["transient_sigmas"] = raw_f[:, 3 + C + 4, :]
            ret["beta"] = beta
        if cfg.nerfh_nff and feat0 is not None:
            ret["feat0"] = feat0
    return ret


def render_rays(ray_batch, network_fn, network_query_fn=None, N_samples=64, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                i_epoch=-1, embedding_a=None, embedding_t=None, test_time=False, args=None, volume=None):
    """Reference signature (rendering.py:68-86) on a packed [n, 8+3(+hist)] ray batch: columns 0:3 origins, 3:6
    directions, 6:8 per-ray near/far (read on the device, no host sync), 8:11 view directions, 11: histogram (unused by
    the field, as in the reference where `ts` never reaches NeRFH_NFF.forward)."""
    if ray_batch.shape[-1] < 11:
        raise NotImplementedError("nefes_amd: the NeFeS field always uses view directions (ray batch needs >= 11 columns)")
    ray_batch = ray_batch.to(_DEV, torch.float32)
    rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
    viewdirs = ray_batch[:, 8:11].contiguous()
    cfg = _cfg(dict(N_samples=N_samples, N_importance=N_importance, perturb=perturb, lindisp=lindisp,
                    white_bkgd=white_bkgd, raw_noise_std=raw_noise_std, test_time=test_time, args=args))
    return _render_core(rays_o, rays_d, viewdirs, 0., 0., network_fn, network_fine, cfg, bounds=ray_batch[:, 6:8])


def _cat_parts(outs):
    return outs[0] if len(outs) == 1 else {k: (torch.cat([o[k] for o in outs], 0) if torch.is_tensor(outs[0][k]) else outs[0][k])
                                           for k in outs[0]}       # (non-tensor entries: flags such as "feat_is_gmap")


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """Reference signature (rendering.py:182-195).  The reference's `chunk` exists to bound memory; here that bound is
    planned against the device memory (rays_per_launch) and the batch is split only when it would not fit."""
    cfg = _cfg(dict(kwargs, N_samples=kwargs.get("N_samples", 64)))
    rays_flat = rays_flat.to(_DEV)
    step = rays_per_launch(cfg, kwargs["network_fn"], kwargs.get("network_fine", None), rays_flat.device,
                           train=_trainable(kwargs["network_fn"]) or _trainable(kwargs.get("network_fine", None)))
    return _cat_parts([render_rays(rays_flat[i:i + step], **kwargs) for i in range(0, rays_flat.shape[0], step)])


def _rays_for(H, W, focal, c2w, c2w_staticcam, row_range):
    row0, nrows = (0, H) if row_range is None else row_range
    rays_o, rays_d, viewdirs = ops.RayGen.apply(c2w.to(_DEV), H, W, float(focal), row0, nrows)
    if c2w_staticcam is not None:                                   # rays of the static camera, view directions of c2w (:211-216)
        rays_o, rays_d, _ = ops.RayGen.apply(c2w_staticcam.to(_DEV), H, W, float(focal), row0, nrows)
    return rays_o, rays_d, viewdirs


def _render_batch(rays_o, rays_d, viewdirs, near, far, chunk, kwargs, cfg):
    network_fn, network_fine = kwargs["network_fn"], kwargs.get("network_fine", None)
    N = rays_o.shape[0]
    # the same predicate _render_core dispatches on: a validation render under torch.no_grad() with trainable networks
    # (run_nefes.py:427,467 -> render_path, rendering.py:350,588) saves nothing and is sliced like any inference batch
    trains = _trainable(network_fn) or _trainable(network_fine)
    step = rays_per_launch(cfg, network_fn, network_fine, rays_o.device, train=trains)
    if not trains and cfg.perturb == 0. and cfg.raw_noise_std == 0.:
        step = _clamp_to_free_memory(step, cfg, network_fn, network_fine, rays_o.device)
    if trains and N > step:
        # every slice's saved activations live until backward(): slicing would not bound the peak, so say it instead of dying later
        raise RuntimeError(f"nefes_amd: a train-mode batch of {N} rays needs more than {MEMORY_FRACTION:.0%} of the device memory for "
                           f"its saved activations (at most {step} rays at this network shape); render fewer rays per step")
    if N <= step:       # the usual case: no slicing (a slice of a differentiable tensor costs a fill and a copy in its backward)
        return _render_core(rays_o, rays_d, viewdirs, float(near), float(far), network_fn, network_fine, cfg)
    outs = []
    for i in range(0, N, step):
        sl = slice(i, min(N, i + step))
        outs.append(_render_core(rays_o[sl], rays_d[sl], viewdirs[sl], float(near), float(far), network_fn, network_fine, cfg))
    return _cat_parts(outs)


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, img_idx=torch.Tensor(0), row_range=None, **kwargs):
    """Reference signature (rendering.py:197-200) + `row_range=(row0, nrows)` for ray-batch sharding.
    Returns [rgb_map [N,3], disp_map [N], acc_map [N], extras dict]."""
    if not use_viewdirs:
        raise NotImplementedError("nefes_amd: the NeFeS field always uses view directions (use_viewdirs=True)")
    cfg = _cfg(kwargs)
    if c2w is not None:
        rays_o, rays_d, viewdirs = _rays_for(int(H), int(W), focal, c2w, c2w_staticcam, row_range)
    else:
        rays_o, rays_d = rays
        rays_o, rays_d = rays_o.to(_DEV).reshape(-1, 3).float(), rays_d.to(_DEV).reshape(-1, 3).float()
        viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)             # caller-supplied rays: torch glue
        if c2w_staticcam is not None:
            rays_o, rays_d, _ = ops.RayGen.apply(c2w_staticcam.to(_DEV), int(H), int(W), float(focal), 0, int(H))
    if ndc:
        rays_o, rays_d = ops.NdcRays.apply(rays_o, rays_d, H, W, float(focal), 1.)
    all_ret = _render_batch(rays_o, rays_d, viewdirs, near, far, chunk, kwargs, cfg)
    k_extract = ["rgb_map", "disp_map", "acc_map"]
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


def render_poses(H, W, focal, poses, chunk=1024 * 32, ndc=True, near=0., far=1., use_viewdirs=False, **kwargs):
    """Several camera poses in ONE launch sequence (the reference's render_path loops render() per
    pose, rendering.py:270-273): rays of all B poses are generated (one small kernel per pose) and concatenated, then
    every heavy kernel -- coarse field, compositing, sampling, fine field, compositing -- runs once over B*H*W rays.
    poses [B,3,4] (or [B,4,4]).  Returns [rgb [B,H*W,3], disp [B,H*W], acc [B,H*W], extras {k: [B,H*W,...]}];
    differentiable w.r.t. `poses` like render()."""
    if not use_viewdirs:
        raise NotImplementedError("nefes_amd: the NeFeS field always uses view directions (use_viewdirs=True)")
    cfg = _cfg(kwargs)
    H, W = int(H), int(W)
    B = poses.shape[0]
    p34 = poses if tuple(poses.shape[-2:]) == (3, 4) else poses[:, :3, :4]
    parts = [_rays_for(H, W, focal, c, None, None) for c in p34.unbind(0)]      # unbind: one stack in the backward, no fills
    rays_o, rays_d, viewdirs = (torch.cat([p[k] for p in parts], 0) for k in range(3))
    if ndc:
        rays_o, rays_d = ops.NdcRays.apply(rays_o, rays_d, H, W, float(focal), 1.)
    all_ret = _render_batch(rays_o, rays_d, viewdirs, near, far, chunk, kwargs, cfg)
    shp = lambda v: v.reshape(B, H * W, *v.shape[1:]) if torch.is_tensor(v) else v
    k_extract = ["rgb_map", "disp_map", "acc_map"]
    return [shp(all_ret[k]) for k in k_extract] + [{k: shp(v) for k, v in all_ret.items() if k not in k_extract}]


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Reference signature (rendering.py:23).  bins [n, B], weights [n, B-1] -> samples [n, N_samples]."""
    dev = _DEV
    u = None if det else torch.rand(bins.shape[0], N_samples, device=dev)
    _, samples = ops.sample_pdf_merge(bins.to(dev), weights.to(dev), N_samples, u=u, bins_layout=True)
    return samples
