"""torch-facing wrappers of the C ABI (include/nefes_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation on the hot
path is a HIP kernel in libnefes_hip.so.  All tensors are fp32 CUDA(=HIP) tensors; wrappers
check dtype/contiguity/device and raise rather than silently converting on a different path.
"""
import ctypes as C
from typing import Optional

import os

import torch

from . import lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing (bench.py): when TIMERS is a dict, every C-ABI launch is bracketed by HIP events
# recorded on the launch stream (torch's current stream is the stream the kernels are launched on).
TIMERS = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMERS is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()

    def __exit__(self, *exc):
        if TIMERS is not None:
            self.t1.record()
            TIMERS.setdefault(self.name, []).append((self.t0, self.t1))
        return False


# Test tap: when TAP is a dict, the forward passes that save ReLU masks append (masks, N, S, width) under "masks" and the
# hierarchical sampler appends its merged depths under "z_fine" -- so that a test can evaluate the float64 oracle on exactly
# the samples and the ReLU branch pattern the kernels used (tests/branch.py).  Never set on the product path.
TAP = None


def _tap(key, value):
    if TAP is not None:
        TAP.setdefault(key, []).append(value)


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"nefes_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"nefes_amd: `{name}` must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"nefes_amd: `{name}` must be contiguous")
    return C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------
# rays
# ---------------------------------------------------------------------------------------------
def raygen_fwd(H, W, focal, c2w, row0=0, nrows=None):
    nrows = H - row0 if nrows is None else nrows
    c2w = _f32(c2w[:3, :4])
    n = nrows * W
    o = torch.empty(n, 3, device=c2w.device)
    d = torch.empty_like(o)
    v = torch.empty_like(o)
    L.check(L.load().nefes_raygen_fwd(H, W, float(focal), _chk(c2w, "c2w"), row0, nrows, _chk(o, "o"), _chk(d, "d"),
                                      _chk(v, "v"), _stream()), "nefes_raygen_fwd")
    return o, d, v


def raygen_bwd(H, W, focal, c2w, row0, nrows, g_o, g_d, g_v):
    lib = L.load()
    c2w = _f32(c2w[:3, :4])
    n = nrows * W
    ws = torch.empty(lib.nefes_raygen_bwd_workspace(n), dtype=torch.uint8, device=c2w.device)
    g = torch.empty(3, 4, device=c2w.device)
    fix = lambda t: None if t is None else _f32(t)
    g_o, g_d, g_v = fix(g_o), fix(g_d), fix(g_v)
    L.check(lib.nefes_raygen_bwd(H, W, float(focal), _chk(c2w, "c2w"), row0, nrows, _chk(g_o, "g_o"), _chk(g_d, "g_d"),
                                 _chk(g_v, "g_v"), _chk(ws, "ws", torch.uint8), _chk(g, "g_c2w"), _stream()),
            "nefes_raygen_bwd")
    return g


class RayGen(torch.autograd.Function):
    """get_rays + viewdirs (ray_utils.py:5-16, rendering.py:217) with the backward to the pose."""

    @staticmethod
    def forward(ctx, c2w, H, W, focal, row0, nrows):
        o, d, v = raygen_fwd(H, W, focal, c2w, row0, nrows)
        ctx.save_for_backward(c2w)
        ctx.geom = (H, W, focal, row0, nrows)
        return o, d, v

    @staticmethod
    def backward(ctx, g_o, g_d, g_v):
        (c2w,) = ctx.saved_tensors
        H, W, focal, row0, nrows = ctx.geom
        g = raygen_bwd(H, W, focal, c2w, row0, nrows, g_o, g_d, g_v)
        if tuple(c2w.shape) == (3, 4) and c2w.dtype == torch.float32:
            return g, None, None, None, None, None         # (the refinement loop's case: no zero fill + copy for a row that is not there)
        out = torch.zeros_like(c2w)
        out[:3, :4] = g
        return out, None, None, None, None, None


class NdcRays(torch.autograd.Function):
    """ndc_rays (ray_utils.py:27-44)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, H, W, focal, near):
        o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
        oo, od = torch.empty_like(o), torch.empty_like(d)
        L.check(L.load().nefes_ndc_fwd(H, W, float(focal), float(near), o.shape[0], _chk(o, "o"), _chk(d, "d"),
                                       _chk(oo, "oo"), _chk(od, "od"), _stream()), "nefes_ndc_fwd")
        ctx.save_for_backward(o, d)
        ctx.geom = (H, W, focal, near, rays_o.shape)
        return oo.reshape(rays_o.shape), od.reshape(rays_d.shape)

    @staticmethod
    def backward(ctx, g_oo, g_od):
        o, d = ctx.saved_tensors
        H, W, focal, near, shape = ctx.geom
        go, gd = torch.empty_like(o), torch.empty_like(d)
        g_oo = None if g_oo is None else _f32(g_oo).reshape(-1, 3)
        g_od = None if g_od is None else _f32(g_od).reshape(-1, 3)
        L.check(L.load().nefes_ndc_bwd(H, W, float(focal), float(near), o.shape[0], _chk(o, "o"), _chk(d, "d"),
                                       _chk(g_oo, "g_oo"), _chk(g_od, "g_od"), _chk(go, "go"), _chk(gd, "gd"), _stream()),
                "nefes_ndc_bwd")
        return go.reshape(shape), gd.reshape(shape), None, None, None, None


_LINSPACE = {}


def _linspace01(n, device):
    """torch.linspace(0, 1, n) on the device, built once per (n, device): rendering.py:33,95 create it on every call."""
    key = (int(n), str(device))
    t = _LINSPACE.get(key)
    if t is None:
        t = _LINSPACE[key] = torch.linspace(0., 1., steps=int(n), device=device)
    return t


def coarse_depths(N, Nc, near, far, lindisp=False, t_rand=None, device="cuda", bounds=None):
    """rendering.py:95-112.  `bounds` = a float32 CUDA tensor whose rows start with (near, far) per ray (any row stride,
    e.g. columns 6:8 of the reference's packed ray batch); then the scalars are ignored."""
    t = _linspace01(Nc, device)                                    # rendering.py:95 (torch's own two-sided formula)
    z = torch.empty(N, Nc, device=device)
    t_rand = None if t_rand is None else _f32(t_rand)
    if bounds is not None:
        if not (bounds.is_cuda and bounds.dtype == torch.float32 and bounds.dim() == 2 and bounds.shape[0] == N
                and bounds.shape[1] >= 2 and bounds.stride(1) == 1):
            raise RuntimeError("nefes_amd: `bounds` must be a float32 CUDA tensor [N, >=2] with unit column stride")
        L.check(L.load().nefes_coarse_depths_rays(N, Nc, C.c_void_p(bounds.data_ptr()), int(bounds.stride(0)),
                                                  int(bool(lindisp)), _chk(t, "t"), _chk(t_rand, "t_rand"), _chk(z, "z"),
                                                  _stream()), "nefes_coarse_depths_rays")
        return z
    L.check(L.load().nefes_coarse_depths(N, Nc, float(near), float(far), int(bool(lindisp)), _chk(t, "t"),
                                         _chk(t_rand, "t_rand"), _chk(z, "z"), _stream()), "nefes_coarse_depths")
    return z


_ZROW = {}
_ZROW_MAX = 64


def coarse_depth_row(Nc, near, far, lindisp=False, device="cuda"):
    """The ONE row of coarse depths every ray shares when near / far are scalars and nothing is jittered (rendering.py:96-100:
    z_vals = near (1 - t) + far t, expanded to [N_rays, N_samples]): [Nc], computed once per (Nc, near, far, lindisp, device) by the
    same kernel as coarse_depths, so the expanded tensor never exists (field_sigma_row, coarse_sample)."""
    key = (int(Nc), float(near), float(far), bool(lindisp), str(device))
    z = _ZROW.get(key)
    if z is not None:
        _ZROW[key] = _ZROW.pop(key)                # most recently used last
        return z
    z = coarse_depths(1, Nc, near, far, lindisp, None, device=device).reshape(-1)
    # A row first computed while a HIP graph is being captured lives in the graph's private pool and is filled on replay only:
    # it must not be handed to later eager renders, so it is not cached.  Per-image near / far values would grow the
    # cache without limit: the least recently used rows go.
    if not torch.cuda.is_current_stream_capturing():
        _ZROW[key] = z
        while len(_ZROW) > _ZROW_MAX:
            _ZROW.pop(next(iter(_ZROW)))
    return z


def fused_coarse_pass_ok(pk, Nc, Ni, N=0, grid=None):
    """The coarse pass as two launches (sigma-only field kernel on a shared depth row + coarse_sample) instead of four: fp16 two-part
    instances, frequency embedding (or a hash grid the field kernel gathers itself: hashgrid_fused_ok), Nc = 64 / 128 / 256,
    Nc + Ni <= 512 (csrc/sample_pdf.hip coarse_sample_kernel)."""
    enc_ok = pk.xyz_encoding == L.XYZ_FREQ10 if grid is None else hashgrid_fused_ok(pk, grid)
    return (FUSED_COARSE and _h3(pk) and h3_shape(pk) and enc_ok and Nc in (64, 128, 256)
            and Ni > 0 and Nc + Ni <= 512 and N * Nc < (1 << 31) - 256)      # (beyond the 32-bit sample index: the four-launch path)


def field_sigma_row(pk, rays_o, rays_d, z_row, grid=None):
    """sigma [N,1,Nc] of the coarse network along rays whose depths are ONE shared row (no gradient: nerfh_nff.py:192-202).
    grid: an ops.HashGrid whose encoding the kernel evaluates itself (hashgrid_fused_ok)."""
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    N, S = rays_o.shape[0], z_row.numel()
    if N * S >= (1 << 31) - 256:
        raise RuntimeError("nefes_amd: too many samples for one launch of the fp16 two-part kernels (32-bit sample index)")
    raw_t = torch.empty(N, 1, S, device=rays_o.device)
    if grid is not None:
        with _timed("field_fwd[sigma,h3,hashgrid]"):
            L.check(L.load().nefes_field_fwd_h3_hashgrid(pk.desc, _chk(pk.blob, "blob", torch.uint8), grid.desc, _chk(grid.table, "table"), L.FIELD_SIGMA,
                                                         N, S, _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(z_row, "z_row"), 1, None,
                                                         _chk(raw_t, "raw_t"), None, _stream()), "nefes_field_fwd_h3_hashgrid")
        return raw_t
    with _timed("field_fwd[sigma,h3]"):
        L.check(L.load().nefes_field_fwd_h3_zrow(pk.desc, _chk(pk.blob, "blob", torch.uint8), L.FIELD_SIGMA, N, S, _chk(rays_o, "rays_o"),
                                                 _chk(rays_d, "rays_d"), _chk(z_row, "z_row"), None, _chk(raw_t, "raw_t"), None, _stream()),
                "nefes_field_fwd_h3_zrow")
    return raw_t


def coarse_sample(sigma, z, Ni, u=None, want_samples=True, want_weights=False):
    """Compositing variant D + sample_pdf + sort(cat) of the coarse pass in ONE launch (csrc/sample_pdf.hip coarse_sample_kernel;
    nerfh_nff.py:83-89, rendering.py:23-66,132-141).  sigma [N,1,Nc] or [N,Nc]; z [N,Nc], or [Nc] = one row shared by every ray.
    -> (z_fine [N,Nc+Ni], z_samples [N,Ni] or None[, weights [N,Nc]]); bit-identical to composite_fwd(COMP_SIGMA_ONLY) +
    sample_pdf_merge."""
    sigma, z = _f32(sigma), _f32(z)
    N, Nc = sigma.shape[0], sigma.shape[-1]
    dev = sigma.device
    if u is None:
        u = _linspace01(Ni, dev)
    u = _f32(u)
    z_fine = torch.empty(N, Nc + Ni, device=dev)
    z_samples = torch.empty(N, Ni, device=dev) if want_samples else None
    weights = torch.empty(N, Nc, device=dev) if want_weights else None
    with _timed("coarse_sample"):
        L.check(L.load().nefes_coarse_sample(N, Nc, Ni, _chk(sigma, "sigma"), _chk(z, "z"), 1 if z.dim() == 1 else 0, _chk(u, "u"),
                                             1 if u.dim() == 2 else 0, _chk(z_fine, "z_fine"), _chk(z_samples, "z_samples"),
                                             _chk(weights, "weights"), _stream()), "nefes_coarse_sample")
    _tap("z_fine", z_fine)
    _tap("z_samples", z_samples)
    return (z_fine, z_samples, weights) if want_weights else (z_fine, z_samples)


# ---------------------------------------------------------------------------------------------
# field MLP
# ---------------------------------------------------------------------------------------------
class PackedField:
    """Device-resident fragment streams of one NeRFH_NFF network (nefes_pack_weights)."""

    LAYERS_COARSE = [f"xyz_encoding_{i}.0" for i in range(1, 9)] + ["xyz_encoding_final", "dir_encoding.0",
                                                                    "static_sigma.0", "static_rgb.0"]
    LAYERS_FINE = LAYERS_COARSE + ["transient_encoding.0", "transient_encoding.2", "transient_encoding.4",
                                   "transient_sigma.0", "transient_rgb.0", "transient_beta.0"]

    def __init__(self, state_dict, width, feat_dim, has_transient, device, xyz_encoding=0):
        lib = L.load()
        self.desc = L.NefesNetDesc(int(width), int(feat_dim), 1 if has_transient else 0, int(xyz_encoding))
        self.xyz_encoding = int(xyz_encoding)
        self.width, self.feat_dim, self.has_transient = int(width), int(feat_dim), bool(has_transient)
        info = L.NefesBlobInfo()
        L.check(lib.nefes_blob_info(self.desc, info), "nefes_blob_info")
        names = self.LAYERS_FINE if has_transient else self.LAYERS_COARSE
        host = []
        for n in names:
            host.append(state_dict[n + ".weight"].detach().to("cpu", torch.float32).contiguous())
            host.append(state_dict[n + ".bias"].detach().to("cpu", torch.float32).contiguous())
        ptrs = (C.c_void_p * len(host))(*[t.data_ptr() for t in host])
        blob = torch.zeros(info.total_bytes, dtype=torch.uint8)
        L.check(lib.nefes_pack_weights(self.desc, ptrs, len(host), C.c_void_p(blob.data_ptr()), blob.numel()),
                "nefes_pack_weights")
        self.blob = blob.to(device)
        self.info = info
        self.generation = 0          # bumped by repack(); autograd nodes refuse to run backward across it
        self.h3_valid = True         # False once a re-pack WITHOUT the fp16 plan left the fp16 two-part streams stale (REPACK_H3 = False)
        self._map = None

    def repack(self, params):
        """Re-pack every stream on the device from the current parameter values (`params`: (weight, bias) per layer in
        LAYERS order, CUDA tensors): one concatenation + one launch (nefes_pack_device), no host copy, no sync."""
        lib = L.load()
        if self._map is None:
            n = self.info.total_bytes // 2
            host_map = torch.zeros(n, dtype=torch.int32)
            elems = (C.c_int64 * 36)()
            L.check(lib.nefes_pack_map(self.desc, C.c_void_p(host_map.data_ptr()), n, C.cast(elems, C.c_void_p)), "nefes_pack_map")
            self._map = host_map.to(self.blob.device)
            self._elems = list(elems)[:36 if self.has_transient else 24]
            # reductions the fp16 two-part streams need (one exponent per weight matrix, row bounds, bias maxima)
            need = C.c_size_t(0)
            L.check(lib.nefes_pack_h3_plan(self.desc, None, 0, C.byref(need)), "nefes_pack_h3_plan")
            plan = torch.zeros(int(need.value), dtype=torch.int32)
            L.check(lib.nefes_pack_h3_plan(self.desc, C.c_void_p(plan.data_ptr()), plan.numel(), None), "nefes_pack_h3_plan")
            self._plan_jobs = int(plan[0])
            self._plan = plan.to(self.blob.device)
            self._plan_scratch = torch.zeros(32, dtype=torch.int32, device=self.blob.device)
        if [int(p.numel()) for p in params] != self._elems:
            raise RuntimeError("nefes_amd: parameter shapes do not match the packed network description")
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32)
        h3 = REPACK_H3 and self._plan_jobs > 0
        L.check(lib.nefes_pack_device(flat.data_ptr(), flat.numel(), self._map.data_ptr(), self._map.numel(),
                                      self._plan.data_ptr() if h3 else None, self._plan_jobs if h3 else 0,
                                      self._plan_scratch.data_ptr() if h3 else None, self.blob.data_ptr(), _stream()),
                "nefes_pack_device")
        self.generation += 1
        if not h3:
            self.h3_valid = False    # the kernels fall back to the bf16x6 instances for this network from here on

    def h3_byte_ranges(self):
        """[(begin, end)] byte ranges of the fp16 two-part units and their scale tables (refreshed by repack() only with the
        fp16 plan, REPACK_H3)."""
        out = []
        for k in (L.STREAM_FWD_SIGMA_H3, L.STREAM_FWD_FULL_H3, L.STREAM_BWD_FULL_H3, L.STREAM_FWD_STATIC_H3, L.STREAM_BWD_STATIC_H3):
            si = self.info.stream[k]
            if si.n_slabs:
                slab = int(L.load().nefes_stream_slab_bytes(self.desc, k))        # per stream: forward and backward rings differ
                out.append((int(si.slab_off), int(si.slab_off + si.n_slabs * slab)))
                out.append((int(si.bias_off + 4 * si.scale_off), int(si.bias_off + 4 * si.bias_floats)))
        return out

    def check_generation(self, gen):
        if gen != self.generation:
            raise RuntimeError("nefes_amd: the network weights were modified (re-packed) between this forward pass and its "
                               "backward pass")

    def mask_bytes(self, M):
        return L.load().nefes_field_mask_bytes(self.desc, M)

    def n_raw(self, mode):
        return 1 if mode == L.FIELD_SIGMA else (3 + self.feat_dim + (1 if mode == L.FIELD_STATIC else 6))


def field_fwd(pk: PackedField, mode, N, S, rays_o=None, rays_d=None, z=None, pts=None, viewdirs=None, want_masks=False,
              xyz_enc=None):
    dev = pk.blob.device
    if not canonical_shape(pk):
        raise RuntimeError(f"nefes_amd: the {('sigma', 'static', 'full')[mode]} forward of W={pk.width}, f_dim={pk.feat_dim} was routed to "
                           f"the fp32-MFMA instances (NEFES_SPLIT={SPLIT}; or a frozen network's static head with test_time=False), "
                           f"which exist for the canonical shapes only.  Compiled: {COMPILED_SET}")
    raw_t = torch.empty(N, pk.n_raw(mode), S, device=dev)
    masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=dev) if want_masks else None
    with _timed(f"field_fwd[{('sigma', 'static', 'full')[mode]}]"):
      L.check(L.load().nefes_field_fwd(pk.desc, _chk(pk.blob, "blob", torch.uint8), mode, N, S, _chk(rays_o, "rays_o"),
                                     _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(xyz_enc, "xyz_enc"),
                                     _chk(viewdirs, "viewdirs"), _chk(raw_t, "raw_t"), _chk(masks, "masks", torch.int32),
                                     _stream()), "nefes_field_fwd")
    if masks is not None:
        _tap("masks", (masks, N, S, pk.width, mode))
    return raw_t, masks


# Which matrix-core arithmetic the field kernels use where an instance exists (width 256 / C = 16, width 128 / C = 128):
#   "h3"  (default) fp16 two-part split products, three cross terms on v_mfma_f32_32x32x16_f16 (csrc/field_h3.h): fp32-level
#         accuracy at half the matrix-core work of bf16x6; tests/test_gpu_h3.py against the float64 oracle
#   "x6"  bf16x6 split products (csrc/field_x6.h): the round-1 default; also what trainable / re-packed networks use
#   "f32" the plain fp32-MFMA kernels (NEFES_X6=0 selects them too)
SPLIT = os.environ.get("NEFES_SPLIT", "h3")
# repack() also refreshes the fp16 two-part streams on the device (csrc/pack_device.hip h3_scales_kernel); "0": leave them stale and
# let re-packed networks run on the bf16x6 instances (the round-2 behaviour)
REPACK_H3 = os.environ.get("NEFES_REPACK_H3", "1") != "0"
USE_X6 = os.environ.get("NEFES_X6", "1") != "0"
# the coarse pass at test time as two launches instead of four (coarse_depth_row / field_sigma_row / coarse_sample); "0": the separate
# coarse_depths, sigma field, composite D and sample_pdf_merge launches (the tests compare the two bit for bit)
FUSED_COARSE = os.environ.get("NEFES_FUSED_COARSE", "1") != "0"


HEAD_MAX_C = 141          # csrc/layout.h NEFES_HEAD_MAX_C: the larger head class serves 3 + C <= 144
COMPILED_SET = L.COMPILED_SET


def head_class(C):
    """csrc/layout.h nefes_head_class: the rgb+feature head's compile-time shape class (0: 3+C <= 32, 1: 3+C <= 144, -1: none)."""
    return -1 if C < 0 else (0 if 3 + C <= 32 else (1 if 3 + C <= 144 else -1))


def canonical_shape(pk: PackedField):
    """Shapes that ALSO have bf16x6 and fp32-MFMA instances: width 256 / C = 16 (either xyz encoding) and the reference-default
    width 128 / C = 128 (frequency embedding)."""
    return (pk.width == 256 and pk.feat_dim == 16) or (pk.width == 128 and pk.feat_dim == 128 and pk.xyz_encoding == L.XYZ_FREQ10)


def h3_shape(pk: PackedField):
    """Shapes with fp16 two-part instances (csrc/field_fwd_h3.hip nefes_field_fwd_h3): both widths x both head classes with the
    frequency embedding; width 256 / class 0 with an external embedding."""
    cls = head_class(pk.feat_dim)
    if pk.width not in (128, 256) or cls < 0:
        return False
    return True if pk.xyz_encoding == L.XYZ_FREQ10 else (pk.width == 256 and cls == 0)


def x6_supported(pk: PackedField, mode, forward=True):
    """A split-product instance (fp16 two-part or bf16x6) serves this network and mode: sigma-only or full, forward and backward --
    and, on the fp16 two-part instances with the frequency embedding, the static head alone (round 5: a frozen coarse network with
    test_time False, a fine network with NeRFW off)."""
    if mode == L.FIELD_STATIC:
        return static_h3(pk)
    ok = canonical_shape(pk) or (_h3(pk) and h3_shape(pk))
    return ok and (mode == L.FIELD_SIGMA or (mode == L.FIELD_FULL and pk.has_transient))


def static_h3(pk: PackedField):
    """The static-head inference instances of the fp16 two-part kernels apply (csrc/field_fwd_h3.hip H3_STATIC,
    nefes_field_bwd_static_h3): every compiled (width, head class) pair with the frequency embedding."""
    return _h3(pk) and h3_shape(pk) and pk.xyz_encoding == L.XYZ_FREQ10


def _h3(pk):
    """fp16 two-part instances apply: selected, this network's fp16 streams are current, and M fits the kernels' 32-bit index."""
    if SPLIT not in ("h3", "x6", "f32"):
        raise ValueError("nefes_amd.ops.SPLIT must be 'h3', 'x6' or 'f32'")
    return SPLIT == "h3" and pk.h3_valid


def require_instance(pk: PackedField, what):
    """Fail loudly, naming the compiled set, before a launch that no kernel instance serves."""
    if canonical_shape(pk) or (_h3(pk) and h3_shape(pk)):
        return
    raise RuntimeError(f"nefes_amd: no kernel instance serves {what} for W={pk.width}, f_dim={pk.feat_dim}, "
                       f"xyz_encoding={pk.xyz_encoding} with NEFES_SPLIT={SPLIT}"
                       f"{'' if pk.h3_valid else ' (fp16 streams stale: NEFES_REPACK_H3=0)'}.  Compiled: {COMPILED_SET}")


def field_fwd_x6(pk: PackedField, mode, N, S, rays_o=None, rays_d=None, z=None, viewdirs=None, want_masks=False, xyz_enc=None,
                 pts=None):
    """field_fwd on the split-product instances (same outputs, same mask words): fp16 two-part (default) or bf16x6."""
    dev = pk.blob.device
    raw_t = torch.empty(N, pk.n_raw(mode), S, device=dev)
    masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=dev) if want_masks else None
    h3 = _h3(pk) and h3_shape(pk) and N * S < (1 << 31) - 256
    if mode == L.FIELD_STATIC and not h3:
        return field_fwd(pk, mode, N, S, rays_o=rays_o, rays_d=rays_d, z=z, pts=pts, viewdirs=viewdirs, want_masks=want_masks, xyz_enc=xyz_enc)
    if not h3 and not canonical_shape(pk):
        raise RuntimeError(f"nefes_amd: {N * S} samples in one launch exceed the fp16 two-part kernels' 32-bit sample index and "
                           f"W={pk.width}, f_dim={pk.feat_dim} has no other instance; render fewer rays per launch")
    fn = L.load().nefes_field_fwd_h3 if h3 else L.load().nefes_field_fwd_x6
    with _timed(f"field_fwd[{('sigma', 'static', 'full')[mode]},{'h3' if h3 else 'x6'}]"):
        L.check(fn(pk.desc, _chk(pk.blob, "blob", torch.uint8), mode, N, S, _chk(rays_o, "rays_o"),
                                            _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(xyz_enc, "xyz_enc"),
                                            _chk(viewdirs, "viewdirs"),
                                            _chk(raw_t, "raw_t"), _chk(masks, "masks", torch.int32), _stream()),
                "nefes_field_fwd_x6")
    if masks is not None:
        _tap("masks", (masks, N, S, pk.width, mode))
    return raw_t, masks


def field_bwd(pk: PackedField, N, S, raw_t, g_raw_t, masks, rays_o=None, rays_d=None, z=None, pts=None, viewdirs=None,
              mode=L.FIELD_FULL):
    dev = pk.blob.device
    ext = pk.xyz_encoding == L.XYZ_EXTERNAL32
    if mode == L.FIELD_STATIC and USE_X6 and SPLIT != "f32" and static_h3(pk) and N * S < (1 << 31) - 256:
        g_pts, g_vs = torch.empty(N * S, 3, device=dev), torch.empty(N * S, 3, device=dev)
        with _timed("field_bwd[static,h3]"):
            L.check(L.load().nefes_field_bwd_static_h3(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                                       _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(viewdirs, "viewdirs"),
                                                       _chk(raw_t, "raw_t"), _chk(g_raw_t, "g_raw_t"), _chk(masks, "masks", torch.int32),
                                                       _chk(g_pts, "g_pts"), _chk(g_vs, "g_vs"), _stream()), "nefes_field_bwd_static_h3")
        return g_pts, g_vs
    if mode == L.FIELD_STATIC:                      # static head only on the fp32 MFMA (canonical shapes)
        if not canonical_shape(pk):
            raise RuntimeError(f"nefes_amd: the static-head backward of W={pk.width}, f_dim={pk.feat_dim} was routed to the fp32-MFMA "
                               f"instances (NEFES_SPLIT={SPLIT}), which exist for the canonical shapes only.  Compiled: {COMPILED_SET}")
        g_pts, g_vs = torch.empty(N * S, 3, device=dev), torch.empty(N * S, 3, device=dev)
        with _timed("field_bwd[static]"):
            L.check(L.load().nefes_field_bwd_static(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                                    _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(viewdirs, "viewdirs"),
                                                    _chk(raw_t, "raw_t"), _chk(g_raw_t, "g_raw_t"), _chk(masks, "masks", torch.int32),
                                                    _chk(g_pts, "g_pts"), _chk(g_vs, "g_vs"), _stream()), "nefes_field_bwd_static")
        return g_pts, g_vs
    g_pts = None if ext else torch.empty(N * S, 3, device=dev)
    g_enc = torch.empty(N * S, 32, device=dev) if ext else None
    g_vs = torch.empty(N * S, 3, device=dev)
    if USE_X6 and SPLIT != "f32" and x6_supported(pk, L.FIELD_FULL, forward=False):
        h3 = _h3(pk)
        fn = L.load().nefes_field_bwd_h3 if h3 else L.load().nefes_field_bwd_x6
        with _timed("field_bwd[h3]" if h3 else "field_bwd[x6]"):
            L.check(fn(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                                _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(viewdirs, "viewdirs"),
                                                _chk(raw_t, "raw_t"), _chk(g_raw_t, "g_raw_t"), _chk(masks, "masks", torch.int32),
                                                _chk(g_pts, "g_pts"), _chk(g_enc, "g_enc"), _chk(g_vs, "g_vs"), _stream()),
                    "nefes_field_bwd_x6")
        return (g_enc if ext else g_pts), g_vs
    with _timed("field_bwd"):
      L.check(L.load().nefes_field_bwd(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                     _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(pts, "pts"), _chk(viewdirs, "viewdirs"),
                                     _chk(raw_t, "raw_t"), _chk(g_raw_t, "g_raw_t"), _chk(masks, "masks", torch.int32),
                                     _chk(g_pts, "g_pts"), _chk(g_enc, "g_enc"), _chk(g_vs, "g_vs"), _stream()),
              "nefes_field_bwd")
    return (g_enc if ext else g_pts), g_vs


def ray_grad_reduce(N, S, z, g_pts, g_vs):
    dev = g_pts.device
    g_o, g_d, g_v = (torch.empty(N, 3, device=dev) for _ in range(3))
    with _timed("ray_grad_reduce"):
      L.check(L.load().nefes_ray_grad_reduce(N, S, _chk(z, "z"), _chk(g_pts, "g_pts"), _chk(g_vs, "g_vs"), _chk(g_o, "g_o"),
                                           _chk(g_d, "g_d"), _chk(g_v, "g_v"), _stream()), "nefes_ray_grad_reduce")
    return g_o, g_d, g_v


class FieldFromRays(torch.autograd.Function):
    """Fused pts = o + d*z -> embed -> MLP (rendering.py:142 + nerfh_nff.py:217-231, :525-576).
    Returns raw_t [N, R, S].  Differentiable w.r.t. rays_o, rays_d, viewdirs in FULL mode (frozen weights)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, z, pk, mode):
        rays_o, rays_d, viewdirs, z = _f32(rays_o), _f32(rays_d), _f32(viewdirs), _f32(z)
        N, S = z.shape
        need = mode in (L.FIELD_FULL, L.FIELD_STATIC) and any(ctx.needs_input_grad[:3])
        if USE_X6 and SPLIT != "f32" and x6_supported(pk, mode):
            raw_t, masks = field_fwd_x6(pk, mode, N, S, rays_o, rays_d, z, viewdirs=viewdirs, want_masks=need)
        else:
            raw_t, masks = field_fwd(pk, mode, N, S, rays_o=rays_o, rays_d=rays_d, z=z, viewdirs=viewdirs, want_masks=need)
        ctx.pk, ctx.mode, ctx.have, ctx.pk_gen = pk, mode, need, pk.generation
        if need:
            ctx.save_for_backward(rays_o, rays_d, viewdirs, z, raw_t, masks)
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        if not ctx.have:
            raise NotImplementedError("nefes_amd: the sigma-only field pass has no backward (the reference evaluates it "
                                      "without gradients at test time: nerfh_nff.py:192-202)")
        rays_o, rays_d, viewdirs, z, raw_t, masks = ctx.saved_tensors
        N, S = z.shape
        ctx.pk.check_generation(ctx.pk_gen)
        g_pts, g_vs = field_bwd(ctx.pk, N, S, raw_t, _f32(g_raw_t), masks, rays_o=rays_o, rays_d=rays_d, z=z,
                                viewdirs=viewdirs, mode=ctx.mode)
        g_o, g_d, g_v = ray_grad_reduce(N, S, z, g_pts, g_vs)
        return g_o, g_d, g_v, None, None, None


# The factored feature head (csrc/field_fwd_h3.hip FH, nefes_amd/render.py): a frozen width-128 fine network emits g = relu(dir_encoding)
# instead of its 128 feature channels and the head's matrix is applied once per ray to the composited g; "0": the plain kernels.
FACTORED_HEAD = os.environ.get("NEFES_FACTORED_HEAD", "1") != "0"


class FeatHead(torch.autograd.Function):
    """The factored head's per-ray part: feat [N, C] = gmap[:, :F] W^T + gmap[:, F:] b (csrc/refine.hip feat_head_*_kernel; W, b frozen)."""

    @staticmethod
    def forward(ctx, gmap, w, w_t, b):
        gmap = _f32(gmap)
        N, F1 = gmap.shape
        C = w.shape[0]
        feat = torch.empty(N, C, device=gmap.device)
        with _timed("feat_head_fwd"):
            L.check(L.load().nefes_feat_head_fwd(N, C, F1 - 1, _chk(gmap, "gmap"), _chk(w_t, "w_t"), _chk(b, "b"), _chk(feat, "feat"), _stream()),
                    "nefes_feat_head_fwd")
        ctx.save_for_backward(w, b)
        ctx.F1 = F1
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        w, b = ctx.saved_tensors
        g_feat = _f32(g_feat)
        N, C = g_feat.shape
        g_gmap = torch.empty(N, ctx.F1, device=g_feat.device)
        with _timed("feat_head_bwd"):
            L.check(L.load().nefes_feat_head_bwd(N, C, ctx.F1 - 1, _chk(g_feat, "g_feat"), _chk(w, "w"), _chk(b, "b"), _chk(g_gmap, "g_gmap"),
                                                 _stream()), "nefes_feat_head_bwd")
        return g_gmap, None, None, None


class FieldFromRaysFH(torch.autograd.Function):
    """FieldFromRays for the factored head: raw_t [N, 3 + (W/2 + 1) + 6, S] = rgb | g | ones | sigma | transient (5); pk = the network packed
    without its feature rows (NeRFH_NFF.packed_fh).  Differentiable w.r.t. rays_o, rays_d, viewdirs (frozen weights)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, z, pk):
        rays_o, rays_d, viewdirs, z = _f32(rays_o), _f32(rays_d), _f32(viewdirs), _f32(z)
        N, S = z.shape
        if N * S >= (1 << 31) - 256:
            raise RuntimeError("nefes_amd: too many samples for one launch of the fp16 two-part kernels (32-bit sample index)")
        need = any(ctx.needs_input_grad[:3])
        R = 3 + pk.width // 2 + 1 + 6
        raw_t = torch.empty(N, R, S, device=z.device)
        masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=z.device) if need else None
        with _timed("field_fwd[full,h3,fh]"):
            L.check(L.load().nefes_field_fwd_h3_fh(pk.desc, _chk(pk.blob, "blob", torch.uint8), L.FIELD_FULL, N, S, _chk(rays_o, "rays_o"),
                                                   _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(viewdirs, "viewdirs"), _chk(raw_t, "raw_t"),
                                                   _chk(masks, "masks", torch.int32), _stream()), "nefes_field_fwd_h3_fh")
        if masks is not None:
            _tap("masks", (masks, N, S, pk.width, L.FIELD_FULL))
        ctx.pk, ctx.have, ctx.pk_gen = pk, need, pk.generation
        if need:
            ctx.save_for_backward(rays_o, rays_d, viewdirs, z, raw_t, masks)
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        if not ctx.have:
            return None, None, None, None, None
        rays_o, rays_d, viewdirs, z, raw_t, masks = ctx.saved_tensors
        N, S = z.shape
        pk = ctx.pk
        pk.check_generation(ctx.pk_gen)
        g_pts, g_vs = torch.empty(N * S, 3, device=z.device), torch.empty(N * S, 3, device=z.device)
        with _timed("field_bwd[h3,fh]"):
            L.check(L.load().nefes_field_bwd_h3_fh(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                                   _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(viewdirs, "viewdirs"), _chk(raw_t, "raw_t"),
                                                   _chk(_f32(g_raw_t), "g_raw_t"), None, _chk(masks, "masks", torch.int32), _chk(g_pts, "g_pts"),
                                                   _chk(g_vs, "g_vs"), _stream()), "nefes_field_bwd_h3_fh")
        g_o, g_d, g_v = ray_grad_reduce(N, S, z, g_pts, g_vs)
        return g_o, g_d, g_v, None, None


class RenderFineFH(torch.autograd.Function):
    """The fine pass of a frozen width-128 network at test time with the factored head, as ONE autograd node: field forward (g channels) ->
    compositing -> per-ray feature head; and back: feature head -> compositing backward, which leaves the static weight where the g
    channels' gradient would go (COMP_FEAT_WEIGHTS_ONLY) -> field backward, which forms d loss / d g = w_s g_gmap[ray] itself.  One node
    because that backward hands TWO tensors from the compositor's stage to the field's (g_raw_t and the per-ray g_gmap).
    -> (rgb [N,3], feat [N,C], disp [N], acc [N]); differentiable w.r.t. rays_o, rays_d, viewdirs."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, z, pk, w_f, w_f_t, b_f, flags, beta_min, emit_gmap=False):
        rays_o, rays_d, viewdirs, z = _f32(rays_o), _f32(rays_d), _f32(viewdirs), _f32(z)
        N, S = z.shape
        need = any(ctx.needs_input_grad[:3])
        Cg = pk.width // 2
        raw_t = torch.empty(N, 3 + Cg + 1 + 6, S, device=z.device)
        masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=z.device) if need else None
        with _timed("field_fwd[full,h3,fh]"):
            L.check(L.load().nefes_field_fwd_h3_fh(pk.desc, _chk(pk.blob, "blob", torch.uint8), L.FIELD_FULL, N, S, _chk(rays_o, "rays_o"),
                                                   _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(viewdirs, "viewdirs"), _chk(raw_t, "raw_t"),
                                                   _chk(masks, "masks", torch.int32), _stream()), "nefes_field_fwd_h3_fh")
        if masks is not None:
            _tap("masks", (masks, N, S, pk.width, L.FIELD_FULL))
        rgb, gmap, disp, acc, _, _, _ = composite_fwd(raw_t, z, Cg + 1, flags, beta_min)
        if emit_gmap:
            # the caller applies the feature head itself, folded into whatever linear layer consumes the features (the refinement loop:
            # FusionNet's first convolution, FusionNet.forward_prepared_gmap): feat = [N, Cg + 1] = (sum_s w_s g_s, sum_s w_s)
            feat = gmap
        else:
            C = w_f.shape[0]
            feat = torch.empty(N, C, device=z.device)
            with _timed("feat_head_fwd"):
                L.check(L.load().nefes_feat_head_fwd(N, C, Cg, _chk(gmap, "gmap"), _chk(w_f_t, "w_t"), _chk(b_f, "b"), _chk(feat, "feat"), _stream()),
                        "nefes_feat_head_fwd")
        ctx.set_materialize_grads(False)
        ctx.pk, ctx.have, ctx.pk_gen, ctx.cfg, ctx.emit_gmap = pk, need, pk.generation, (Cg, int(flags)), bool(emit_gmap)
        if need:
            ctx.save_for_backward(rays_o, rays_d, viewdirs, z, raw_t, masks, w_f, b_f)
        return rgb, feat, disp, acc

    @staticmethod
    def backward(ctx, g_rgb, g_feat, g_disp, g_acc):
        if not ctx.have:
            return (None,) * 11
        rays_o, rays_d, viewdirs, z, raw_t, masks, w_f, b_f = ctx.saved_tensors
        N, S = z.shape
        pk = ctx.pk
        pk.check_generation(ctx.pk_gen)
        Cg, flags = ctx.cfg
        fix = lambda g: None if (g is None or g.numel() == 0) else _f32(g)
        g_rgb, g_feat, g_disp, g_acc = fix(g_rgb), fix(g_feat), fix(g_disp), fix(g_acc)
        g_gmap = None
        if g_feat is not None and ctx.emit_gmap:
            g_gmap = g_feat                                  # already d loss / d (sum_s w_s g_s, sum_s w_s)
        elif g_feat is not None:
            g_gmap = torch.empty(N, Cg + 1, device=z.device)
            with _timed("feat_head_bwd"):
                L.check(L.load().nefes_feat_head_bwd(N, w_f.shape[0], Cg, _chk(g_feat, "g_feat"), _chk(w_f, "w"), _chk(b_f, "b"),
                                                     _chk(g_gmap, "g_gmap"), _stream()), "nefes_feat_head_bwd")
        else:
            g_gmap = torch.zeros(N, Cg + 1, device=z.device)
        g_raw_t = torch.empty_like(raw_t)          # (the g channels' rows beyond the first are neither written nor read)
        with _timed("composite_bwd"):
            L.check(L.load().nefes_composite_bwd(N, S, Cg + 1, flags | L.COMP_FEAT_WEIGHTS_ONLY, _chk(raw_t, "raw_t"), _chk(z, "z"),
                                                 _chk(g_rgb, "g_rgb"), None, _chk(g_disp, "g_disp"), _chk(g_acc, "g_acc"), None, None, None,
                                                 _chk(g_raw_t, "g_raw_t"), _stream()), "nefes_composite_bwd")
        g_pts, g_vs = torch.empty(N * S, 3, device=z.device), torch.empty(N * S, 3, device=z.device)
        with _timed("field_bwd[h3,fh]"):
            L.check(L.load().nefes_field_bwd_h3_fh(pk.desc, _chk(pk.blob, "blob", torch.uint8), N, S, _chk(rays_o, "rays_o"),
                                                   _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(viewdirs, "viewdirs"), _chk(raw_t, "raw_t"),
                                                   _chk(g_raw_t, "g_raw_t"), _chk(g_gmap, "g_gmap"), _chk(masks, "masks", torch.int32),
                                                   _chk(g_pts, "g_pts"), _chk(g_vs, "g_vs"), _stream()), "nefes_field_bwd_h3_fh")
        g_o, g_d, g_v = ray_grad_reduce(N, S, z, g_pts, g_vs)
        return g_o, g_d, g_v, None, None, None, None, None, None, None, None


class FieldFromPoints(torch.autograd.Function):
    """run_network_NeRFH_NFF call surface: explicit pts [N,S,3] (+ viewdirs [N,3]) -> raw_t [N,R,S]."""

    @staticmethod
    def forward(ctx, pts, viewdirs, pk, mode):
        pts = _f32(pts)
        N, S = pts.shape[0], pts.shape[1]
        if viewdirs is None:
            viewdirs = torch.zeros(N, 3, device=pts.device)
        viewdirs = _f32(viewdirs)
        need = mode in (L.FIELD_FULL, L.FIELD_STATIC) and any(ctx.needs_input_grad[:2])
        if USE_X6 and SPLIT != "f32" and x6_supported(pk, mode) and pk.xyz_encoding == L.XYZ_FREQ10:
            raw_t, masks = field_fwd_x6(pk, mode, N, S, pts=pts.reshape(-1, 3), viewdirs=viewdirs, want_masks=need)
        else:
            raw_t, masks = field_fwd(pk, mode, N, S, pts=pts.reshape(-1, 3), viewdirs=viewdirs, want_masks=need)
        ctx.pk, ctx.mode, ctx.have, ctx.pk_gen = pk, mode, need, pk.generation
        if need:
            ctx.save_for_backward(pts, viewdirs, raw_t, masks)
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        if not ctx.have:
            raise NotImplementedError("nefes_amd: the sigma-only field pass has no backward (nerfh_nff.py:192-202 runs it without gradients)")
        pts, viewdirs, raw_t, masks = ctx.saved_tensors
        N, S = pts.shape[0], pts.shape[1]
        ctx.pk.check_generation(ctx.pk_gen)
        g_pts, g_vs = field_bwd(ctx.pk, N, S, raw_t, _f32(g_raw_t), masks, pts=pts.reshape(-1, 3), viewdirs=viewdirs, mode=ctx.mode)
        zeros = torch.zeros(N, S, device=pts.device)
        _, _, g_v = ray_grad_reduce(N, S, zeros, g_pts, g_vs)
        return g_pts.reshape(N, S, 3), g_v, None, None


class FieldFromEncoding(torch.autograd.Function):
    """Field MLP on a caller-supplied 32-feature xyz embedding (hash grid, BASELINE config 4): enc [N,S,32], viewdirs [N,3]
    -> raw_t [N,R,S]; backward to enc and viewdirs."""

    @staticmethod
    def forward(ctx, enc, viewdirs, pk, mode):
        enc = _f32(enc)
        N, S = enc.shape[0], enc.shape[1]
        viewdirs = torch.zeros(N, 3, device=enc.device) if viewdirs is None else _f32(viewdirs)
        need = mode == L.FIELD_FULL and any(ctx.needs_input_grad[:2])
        if USE_X6 and SPLIT != "f32" and x6_supported(pk, mode):
            raw_t, masks = field_fwd_x6(pk, mode, N, S, xyz_enc=enc.reshape(-1, 32), viewdirs=viewdirs, want_masks=need)
        else:
            raw_t, masks = field_fwd(pk, mode, N, S, xyz_enc=enc.reshape(-1, 32), viewdirs=viewdirs, want_masks=need)
        ctx.pk, ctx.have, ctx.pk_gen = pk, need, pk.generation
        if need:
            ctx.save_for_backward(viewdirs, raw_t, masks)
            ctx.shape = (N, S)
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        if not ctx.have:
            raise NotImplementedError("nefes_amd: field backward is built for the FULL (fine) mode only")
        viewdirs, raw_t, masks = ctx.saved_tensors
        N, S = ctx.shape
        ctx.pk.check_generation(ctx.pk_gen)
        g_enc, g_vs = field_bwd(ctx.pk, N, S, raw_t, _f32(g_raw_t), masks, viewdirs=viewdirs)
        zeros = torch.zeros(N, S, device=raw_t.device)
        _, _, g_v = ray_grad_reduce(N, S, zeros, g_vs, g_vs)
        return g_enc.reshape(N, S, 32), g_v, None, None


# ---------------------------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------------------------
def composite_fwd(raw_t, z, C_feat, flags, beta_min=0.1):
    N, _, S = raw_t.shape
    dev = raw_t.device
    sigma_only = bool(flags & L.COMP_SIGMA_ONLY)
    acc = torch.empty(N, device=dev)
    weights = torch.empty(N, S, device=dev)
    if sigma_only:
        rgb = feat = disp = depth = beta = None
    else:
        rgb, feat = torch.empty(N, 3, device=dev), torch.empty(N, C_feat, device=dev)
        disp, depth, beta = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    with _timed("composite_fwd[D]" if sigma_only else "composite_fwd"):
      L.check(L.load().nefes_composite_fwd(N, S, C_feat, flags, float(beta_min), _chk(raw_t, "raw_t"), _chk(z, "z"),
                                         _chk(rgb, "rgb"), _chk(feat, "feat"), _chk(disp, "disp"), _chk(acc, "acc"),
                                         _chk(depth, "depth"), _chk(weights, "weights"), _chk(beta, "beta"), _stream()),
            "nefes_composite_fwd")
    return rgb, feat, disp, acc, depth, weights, beta


class Composite(torch.autograd.Function):
    """raw2outputs_NeRFH_NFF (nerfh_nff.py:25-166) on raw_t [N,R,S]; z carries no gradient (detached in the reference)."""

    @staticmethod
    def forward(ctx, raw_t, z, C_feat, flags, beta_min):
        raw_t, z = _f32(raw_t), _f32(z)
        outs = composite_fwd(raw_t, z, C_feat, flags, beta_min)
        ctx.set_materialize_grads(False)      # maps nobody differentiates arrive as None, not as freshly zero-filled tensors
        ctx.save_for_backward(raw_t, z)
        ctx.cfg = (C_feat, flags)
        if flags & L.COMP_SIGMA_ONLY:
            _, _, _, acc, _, weights, _ = outs
            dummy = raw_t.new_zeros(0)
            return dummy, dummy, dummy, acc, dummy, weights, dummy
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_feat, g_disp, g_acc, g_depth, g_weights, g_beta):
        raw_t, z = ctx.saved_tensors
        C_feat, flags = ctx.cfg
        N, R, S = raw_t.shape
        g_raw_t = torch.empty_like(raw_t)
        so = bool(flags & L.COMP_SIGMA_ONLY)
        fix = lambda g, ok=True: None if (g is None or not ok or g.numel() == 0) else _f32(g)
        g_rgb, g_feat, g_disp, g_depth, g_beta = (fix(g, not so) for g in (g_rgb, g_feat, g_disp, g_depth, g_beta))
        g_acc, g_weights = fix(g_acc), fix(g_weights)
        with _timed("composite_bwd"):
          L.check(L.load().nefes_composite_bwd(N, S, C_feat, flags, _chk(raw_t, "raw_t"), _chk(z, "z"), _chk(g_rgb, "g_rgb"),
                                             _chk(g_feat, "g_feat"), _chk(g_disp, "g_disp"), _chk(g_acc, "g_acc"),
                                             _chk(g_depth, "g_depth"), _chk(g_weights, "g_weights"), _chk(g_beta, "g_beta"),
                                             _chk(g_raw_t, "g_raw_t"), _stream()), "nefes_composite_bwd")
        return g_raw_t, None, None, None, None


# ---------------------------------------------------------------------------------------------
# hierarchical sampling
# ---------------------------------------------------------------------------------------------
def sample_pdf_merge(z_coarse, weights, Ni, u=None, cdf=None, want_debug=False, bins_layout=False):
    """sample_pdf + sort(cat) (rendering.py:23-66,132-141).  u=None -> deterministic linspace(0,1,Ni).
    bins_layout=True: (z_coarse, weights) are the reference's (bins, weights) arguments; no merge."""
    z_coarse, weights = _f32(z_coarse), _f32(weights)
    N, Nc = z_coarse.shape
    if bins_layout:
        Nc += 1
    dev = z_coarse.device
    if u is None:
        u = _linspace01(Ni, dev)                                     # rendering.py:33 bits
    u = _f32(u)
    per_ray = 1 if u.dim() == 2 else 0
    z_fine = None if bins_layout else torch.empty(N, Nc + Ni, device=dev)
    z_samples = torch.empty(N, Ni, device=dev)
    inds = torch.empty(N, Ni, dtype=torch.int32, device=dev) if want_debug else None
    cdf_out = torch.empty(N, Nc - 1, device=dev) if want_debug else None
    cdf = None if cdf is None else _f32(cdf)
    with _timed("sample_pdf_merge"):
      L.check(L.load().nefes_sample_pdf_merge(N, Nc, Ni, 1 if bins_layout else 0, _chk(z_coarse, "z_coarse"), _chk(weights, "weights"), _chk(u, "u"),
                                            per_ray, _chk(cdf, "cdf"), _chk(z_fine, "z_fine"), _chk(z_samples, "z_samples"),
                                            _chk(inds, "inds", torch.int32), _chk(cdf_out, "cdf_out"), _stream()),
            "nefes_sample_pdf_merge")
    _tap("z_fine", z_fine)
    _tap("z_samples", z_samples)
    if want_debug:
        return z_fine, z_samples, inds, cdf_out
    return z_fine, z_samples


# ---------------------------------------------------------------------------------------------
# hash-grid encoding (BASELINE config 4; parity unpinned, see oracle/hashgrid_ref.py)
# ---------------------------------------------------------------------------------------------
class HashGrid:
    """Table + geometry of a tiny-cuda-nn style multiresolution hash grid (nerfh_tcnn.py:60-75)."""

    def __init__(self, bound, table=None, n_levels=16, log2_hashmap_size=19, base_resolution=16, max_resolution=2048,
                 device="cuda"):
        import math
        self.desc = L.NefesHashGridDesc(n_levels, 2, log2_hashmap_size, base_resolution,
                                        math.exp(math.log(max_resolution / base_resolution) / (n_levels - 1)), float(bound))
        self.n_out = 2 * n_levels
        n = L.load().nefes_hashgrid_table_entries(self.desc)
        if n == 0:
            raise RuntimeError("nefes_amd: unsupported hash-grid configuration")
        if table is None:
            g = torch.Generator().manual_seed(0)
            table = (torch.rand(n, 2, generator=g) * 2 - 1) * 1e-4
        if tuple(table.shape) != (n, 2):
            raise RuntimeError(f"nefes_amd: hash-grid table must be [{n}, 2]")
        self.table = table.to(device, torch.float32).contiguous()

    def __call__(self, x):
        return HashGridEncode.apply(x, self)


# BASELINE configs[3] with the hash grid evaluated INSIDE the field kernels (csrc/hashgrid.h, nefes_field_fwd_h3_hashgrid /
# nefes_field_bwd_h3_hashgrid); "0": the separate launches HashGridEncode + FieldFromEncoding (the tests compare the two)
FUSED_HASHGRID = os.environ.get("NEFES_FUSED_HASHGRID", "1") != "0"


def hashgrid_fused_ok(pk, grid):
    """The fp16 two-part field kernels can gather this hash grid themselves: width 256, head class 0, sixteen levels x two features."""
    return (FUSED_HASHGRID and isinstance(grid, HashGrid) and _h3(pk) and pk.xyz_encoding == L.XYZ_EXTERNAL32 and pk.width == 256
            and head_class(pk.feat_dim) == 0 and grid.desc.n_levels == 16 and grid.desc.n_features == 2)


class FieldFromRaysHashGrid(torch.autograd.Function):
    """pts = o + d z -> hash-grid encoding -> MLP in ONE launch each way (rendering.py:114,142 + nerfh_tcnn.py:151-182): raw_t [N, R, S];
    differentiable w.r.t. rays_o, rays_d, viewdirs in FULL mode (frozen weights, frozen table).  Forward bit-identical to
    HashGridEncode + FieldFromEncoding."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, z, pk, mode, grid):
        rays_o, rays_d, viewdirs, z = _f32(rays_o), _f32(rays_d), _f32(viewdirs), _f32(z)
        N, S = z.shape
        if N * S >= (1 << 31) - 256:
            raise RuntimeError("nefes_amd: too many samples for one launch of the fp16 two-part kernels (32-bit sample index)")
        if mode not in (L.FIELD_SIGMA, L.FIELD_FULL):
            raise NotImplementedError("nefes_amd: the hash-grid field kernels evaluate the sigma-only or the full head")
        need = mode == L.FIELD_FULL and any(ctx.needs_input_grad[:3])
        raw_t = torch.empty(N, pk.n_raw(mode), S, device=z.device)
        masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=z.device) if need else None
        with _timed(f"field_fwd[{('sigma', 'static', 'full')[mode]},h3,hashgrid]"):
            L.check(L.load().nefes_field_fwd_h3_hashgrid(pk.desc, _chk(pk.blob, "blob", torch.uint8), grid.desc, _chk(grid.table, "table"), mode, N, S,
                                                         _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(z, "z"), 0, _chk(viewdirs, "viewdirs"),
                                                         _chk(raw_t, "raw_t"), _chk(masks, "masks", torch.int32), _stream()),
                    "nefes_field_fwd_h3_hashgrid")
        if masks is not None:
            _tap("masks", (masks, N, S, pk.width, mode))
        ctx.pk, ctx.grid, ctx.have, ctx.pk_gen = pk, grid, need, pk.generation
        if need:
            ctx.save_for_backward(rays_o, rays_d, viewdirs, z, raw_t, masks)
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        if not ctx.have:
            raise NotImplementedError("nefes_amd: the sigma-only field pass has no backward (nerfh_nff.py:192-202 runs it without gradients)")
        rays_o, rays_d, viewdirs, z, raw_t, masks = ctx.saved_tensors
        N, S = z.shape
        pk, grid = ctx.pk, ctx.grid
        pk.check_generation(ctx.pk_gen)
        g_pts, g_vs = torch.empty(N * S, 3, device=z.device), torch.empty(N * S, 3, device=z.device)
        with _timed("field_bwd[h3,hashgrid]"):
            L.check(L.load().nefes_field_bwd_h3_hashgrid(pk.desc, _chk(pk.blob, "blob", torch.uint8), grid.desc, _chk(grid.table, "table"), N, S,
                                                         _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(z, "z"), _chk(viewdirs, "viewdirs"),
                                                         _chk(raw_t, "raw_t"), _chk(_f32(g_raw_t), "g_raw_t"), _chk(masks, "masks", torch.int32),
                                                         _chk(g_pts, "g_pts"), _chk(g_vs, "g_vs"), _stream()), "nefes_field_bwd_h3_hashgrid")
        g_o, g_d, g_v = ray_grad_reduce(N, S, z, g_pts, g_vs)
        return g_o, g_d, g_v, None, None, None, None


class HashGridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid):
        shape = x.shape
        xf = _f32(x).reshape(-1, 3)
        enc = torch.empty(xf.shape[0], grid.n_out, device=xf.device)
        with _timed("hashgrid_fwd"):
            L.check(L.load().nefes_hashgrid_fwd(grid.desc, _chk(grid.table, "table"), xf.shape[0], _chk(xf, "x"),
                                                _chk(enc, "enc"), _stream()), "nefes_hashgrid_fwd")
        ctx.save_for_backward(xf)
        ctx.grid, ctx.shape = grid, shape
        return enc.reshape(*shape[:-1], grid.n_out)

    @staticmethod
    def backward(ctx, g_enc):
        (xf,) = ctx.saved_tensors
        g = _f32(g_enc).reshape(-1, ctx.grid.n_out)
        g_x = torch.empty_like(xf)
        with _timed("hashgrid_bwd_x"):
            L.check(L.load().nefes_hashgrid_bwd_x(ctx.grid.desc, _chk(ctx.grid.table, "table"), xf.shape[0], _chk(xf, "x"),
                                                  _chk(g, "g_enc"), _chk(g_x, "g_x"), _stream()), "nefes_hashgrid_bwd_x")
        return g_x.reshape(ctx.shape), None


_CONV_PACK = {}     # (id(weight), backward) -> (weakref to the weight, its version, packed [Cin even][K*K][Cout multiple of 32])


def _pack_conv(w, backward):
    """Packed A operands of nefes_conv2d_same for a frozen Conv2d weight [Cout, Cin, K, K]; backward: the flipped, transposed
    weights (gradient w.r.t. the layer's input).  Cached per weight tensor and version (the entry holds a weak reference: an
    address re-used by another tensor is not a hit)."""
    import weakref
    key = (id(w), backward)
    hit = _CONV_PACK.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    wd = w.detach().to(torch.float32)
    if backward:
        wd = wd.flip(2, 3).transpose(0, 1)
    co, ci, k, _ = wd.shape
    pk = torch.zeros((ci + 1) // 2 * 2, k * k, (co + 31) // 32 * 32, device=w.device)
    pk[:ci, :, :co] = wd.permute(1, 2, 3, 0).reshape(ci, k * k, co)
    for dead in [kk for kk, v in _CONV_PACK.items() if v[0]() is None]:
        del _CONV_PACK[dead]
    _CONV_PACK[key] = (weakref.ref(w), w._version, pk)
    return pk


def _conv2d_same(x, wp, cout, ksize, bias, relu, mask=None):
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, device=x.device)
    with _timed("conv2d_same"):
        L.check(L.load().nefes_conv2d_same(B, cin, cout, H, W, ksize, _chk(x, "x"), _chk(mask, "mask"), _chk(wp, "w_packed"),
                                           _chk(bias, "bias"), int(relu), y.data_ptr(), _stream()), "nefes_conv2d_same")
    return y


class FrozenConv2d(torch.autograd.Function):
    """Conv2d(stride 1, "same" zero padding, 3x3 or 5x5) [+ ReLU] with FROZEN weight and bias, differentiable w.r.t. its input:
    FusionNet's layers in the refinement loop (nerfh_nff.py:356-418; weights carry no gradient there).  Forward and the input
    gradient are the same implicit-GEMM kernel (csrc/conv.hip); the ReLU derivative is applied to the incoming gradient inside
    the gradient launch (mask = this layer's output)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        xc = _f32(x)
        k = weight.shape[-1]
        y = _conv2d_same(xc, _pack_conv(weight, False), weight.shape[0], k, None if bias is None else _f32(bias.detach()), relu)
        ctx.weight, ctx.relu = weight, relu
        if relu:
            ctx.save_for_backward(y)
            _tap("conv_relu", y)             # tests: the ReLU branch pattern of this layer (y > 0), tests/branch.py's rule for FusionNet
        return y

    @staticmethod
    def backward(ctx, g):
        w = ctx.weight
        y = ctx.saved_tensors[0] if ctx.relu else None
        gx = _conv2d_same(_f32(g), _pack_conv(w, True), w.shape[1], w.shape[-1], None, False, mask=y)
        return gx, None, None, None


def frozen_conv2d(x, weight, bias, relu=False):
    if weight.requires_grad or (bias is not None and bias.requires_grad):
        raise ValueError("nefes_amd: frozen_conv2d is for weights without gradient (use torch's conv2d to train them)")
    return FrozenConv2d.apply(x, weight, bias, relu)


class BicubicUpsample(torch.autograd.Function):
    """torch.nn.Upsample(size=(OH, OW), mode='bicubic') on a contiguous [B,C,h,w] image (DFM_APR_refine.py:114,118), optionally
    only the window `crop` pixels inside every border (the loop's `[:, :, 10:-10, 10:-10]`, :115,119: the cropped-away pixels are
    never computed and the slice's backward -- a 39 MB fill plus a copy -- disappears); the backward is a separable gather
    instead of the library's atomic scatter."""

    @staticmethod
    def forward(ctx, x, OH, OW, crop):
        B, Cc, h, w = x.shape
        xf = _f32(x)
        CH, CW = OH - 2 * crop, OW - 2 * crop
        out = torch.empty(B, Cc, CH, CW, device=xf.device)
        with _timed("bicubic_up_fwd"):
            L.check(L.load().nefes_bicubic_up_fwd(B * Cc, h, w, OH, OW, crop, crop, CH, CW, _chk(xf, "x"), _chk(out, "out"), _stream()),
                    "nefes_bicubic_up_fwd")
        ctx.dims = (B, Cc, h, w, OH, OW, crop, CH, CW)
        return out

    @staticmethod
    def backward(ctx, g_out):
        B, Cc, h, w, OH, OW, crop, CH, CW = ctx.dims
        g = _f32(g_out)
        tmp = torch.empty(B * Cc, h, CW, device=g.device)
        g_in = torch.empty(B, Cc, h, w, device=g.device)
        with _timed("bicubic_up_bwd"):
            L.check(L.load().nefes_bicubic_up_bwd(B * Cc, h, w, OH, OW, crop, crop, CH, CW, _chk(g, "g_out"), _chk(tmp, "tmp"),
                                                  _chk(g_in, "g_in"), _stream()), "nefes_bicubic_up_bwd")
        return g_in, None, None, None


def bicubic_upsample(x, size, crop=0):
    return BicubicUpsample.apply(x, int(size[0]), int(size[1]), int(crop))


class PoseCompose(torch.autograd.Function):
    """LearnPose.forward (models/poses.py:43-50, lietorch=False) + fix_coord_supp (dm/direct_pose_model.py:224-231) in one
    launch each way: (r [n,3], t [n,3]) -> c2w [n,3,4] in NeRF coordinates.  `init_c2w` [n,4,4] on the device, no gradient."""

    @staticmethod
    def forward(ctx, r, t, init_c2w, pose_scale, move, pose_scale2, grad_into=None):
        rf, tf, i0 = _f32(r).reshape(-1, 3), _f32(t).reshape(-1, 3), _f32(init_c2w).reshape(-1, 4, 4)
        ctx.grad_into = grad_into             # (g_r, g_t) buffers the backward writes (the parameters' .grad: no copy afterwards)
        n = rf.shape[0]
        if tf.shape[0] != n or i0.shape[0] != n:
            raise ValueError("nefes_amd: pose_compose needs as many translations and initial poses as rotations")
        mv = (C.c_float * 3)(*[float(v) for v in move])
        out = torch.empty(n, 3, 4, device=rf.device)
        L.check(L.load().nefes_pose_compose_fwd(n, _chk(rf, "r"), _chk(tf, "t"), _chk(i0, "init_c2w"), float(pose_scale), mv,
                                                float(pose_scale2), _chk(out, "c2w"), _stream()), "nefes_pose_compose_fwd")
        ctx.save_for_backward(rf, tf, i0)
        ctx.consts = (float(pose_scale), mv, float(pose_scale2), r.shape, t.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        rf, tf, i0 = ctx.saved_tensors
        sc, mv, sc2, r_shape, t_shape = ctx.consts
        gf = _f32(g)
        if ctx.grad_into is not None:
            g_r, g_t = (b.reshape(rf.shape) for b in ctx.grad_into)
        else:
            g_r, g_t = torch.empty_like(rf), torch.empty_like(tf)
        L.check(L.load().nefes_pose_compose_bwd(rf.shape[0], _chk(rf, "r"), _chk(tf, "t"), _chk(i0, "init_c2w"), sc, mv, sc2,
                                                _chk(gf, "g_c2w"), _chk(g_r, "g_r"), _chk(g_t, "g_t"), _stream()),
                "nefes_pose_compose_bwd")
        return g_r.reshape(r_shape), g_t.reshape(t_shape), None, None, None, None, None


def pose_compose(r, t, init_c2w, pose_scale=1.0, move=(0., 0., 0.), pose_scale2=1.0, grad_into=None):
    """[n,3,4] for n cameras; [3,4] when r is a single 3-vector.  grad_into = (g_r, g_t): buffers the backward writes."""
    out = PoseCompose.apply(r, t, init_c2w, pose_scale, move, pose_scale2, grad_into)
    return out[0] if r.dim() == 1 else out


class BatchNormTrainFrozen(torch.autograd.Function):
    """torch.nn.BatchNorm2d in TRAIN mode with frozen affine parameters on [B,C,H,W] (FusionNet's last layer in the refinement loop,
    nerfh_nff.py:356-418): output normalised by the batch's statistics (running statistics and the batch counter updated as the module
    does), or by every image's own (`per_image`: torch's instance_norm with the BatchNorm's affine parameters, what the reference's
    one-image-at-a-time loop computes; running statistics untouched).  csrc/refine.hip bn_train_*: float64 sums, gradient to x only."""

    @staticmethod
    def forward(ctx, x, bn, per_image, track_stats=True):
        xf = _f32(x)
        B, Cc = xf.shape[0], xf.shape[1]
        P = xf.numel() // (B * Cc)
        y = torch.empty_like(xf)
        groups = B if per_image else 1
        save = torch.empty(groups * Cc * 2, dtype=torch.float64, device=xf.device)
        w = None if bn.weight is None else _f32(bn.weight)
        b = None if bn.bias is None else _f32(bn.bias)
        track = bool(track_stats) and (not per_image) and bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1
        if track:
            if bn.momentum is None:
                raise NotImplementedError("nefes_amd: BatchNorm with momentum=None (cumulative average) takes the torch module")
            momentum = float(bn.momentum)
        L.check(L.load().nefes_bn_train_fwd(B, Cc, P, 1 if per_image else 0, _chk(xf, "x"), _chk(w, "weight"), _chk(b, "bias"), float(bn.eps), momentum,
                                            _chk(bn.running_mean, "running_mean") if track else None, _chk(bn.running_var, "running_var") if track else None,
                                            _chk(bn.num_batches_tracked, "num_batches_tracked", torch.int64) if track and bn.num_batches_tracked is not None else None,
                                            _chk(y, "y"), _chk(save, "save", torch.float64), _stream()), "nefes_bn_train_fwd")
        ctx.save_for_backward(xf, save)
        ctx.w, ctx.per_image = w, bool(per_image)
        return y

    @staticmethod
    def backward(ctx, g):
        xf, save = ctx.saved_tensors
        gf = _f32(g)
        B, Cc = xf.shape[0], xf.shape[1]
        g_x = torch.empty_like(xf)
        L.check(L.load().nefes_bn_train_bwd(B, Cc, xf.numel() // (B * Cc), 1 if ctx.per_image else 0, _chk(xf, "x"), _chk(ctx.w, "weight"),
                                            _chk(save, "save", torch.float64), _chk(gf, "g_y"), _chk(g_x, "g_x"), _stream()), "nefes_bn_train_bwd")
        return g_x, None, None, None


def batch_norm_train_frozen(x, bn, per_image=False, track_stats=True):
    """bn(x) for a torch.nn.BatchNorm2d in train mode whose weight and bias carry no gradient (see BatchNormTrainFrozen).
    track_stats=False: the module's running statistics and batch counter are left alone (the outputs do not depend on them in train
    mode): what a caller passes whose launches run on several streams at once over ONE shared module -- the update is a plain
    read-modify-write of the module's buffers (nefes_amd.refine.refine_concurrently)."""
    if not bn.training or any(p is not None and p.requires_grad for p in (bn.weight, bn.bias)):
        raise ValueError("nefes_amd: batch_norm_train_frozen is for a BatchNorm in train mode with frozen affine parameters")
    return BatchNormTrainFrozen.apply(x, bn, bool(per_image), bool(track_stats))


class SvdReg(torch.autograd.Function):
    """svd_reg (dm/DFM_pose_refine.py:119-129) on [..., 3, 4] poses: rotation block -> U V^T, one launch each way (csrc/refine.hip
    svd_reg_*; float64 inside; the backward is the polar factor's derivative, not autograd through an SVD)."""

    @staticmethod
    def forward(ctx, pose):
        pf = _f32(pose).reshape(-1, 3, 4)
        n = pf.shape[0]
        out = torch.empty_like(pf)
        save = torch.empty(n, 21, dtype=torch.float64, device=pf.device)
        L.check(L.load().nefes_svd_reg_fwd(n, _chk(pf, "pose"), _chk(out, "out"), _chk(save, "save", torch.float64), _stream()), "nefes_svd_reg_fwd")
        ctx.save_for_backward(save)
        ctx.shape = pose.shape
        return out.reshape(pose.shape)

    @staticmethod
    def backward(ctx, g):
        (save,) = ctx.saved_tensors
        gf = _f32(g).reshape(-1, 3, 4)
        g_pose = torch.empty_like(gf)
        L.check(L.load().nefes_svd_reg_bwd(gf.shape[0], _chk(save, "save", torch.float64), _chk(gf, "g_out"), _chk(g_pose, "g_pose"), _stream()),
                "nefes_svd_reg_bwd")
        return g_pose.reshape(ctx.shape)


def svd_reg(pose):
    """[..., 3, 4] poses with the 3x3 block replaced by its nearest orthogonal matrix U V^T (include/nefes_hip.h nefes_svd_reg_fwd)."""
    return SvdReg.apply(pose)


class RegressedPose(torch.autograd.Function):
    """svd_reg (optional) + fix_coord_supp on [..., 3, 4] regressed poses, one launch each way (include/nefes_hip.h
    nefes_regressed_pose_fwd): what train_on_batch does to the regression network's output before it renders (DFM_APR_refine.py:91-97)."""

    @staticmethod
    def forward(ctx, pose, do_svd, t_scale, move):
        pf = _f32(pose).reshape(-1, 3, 4)
        n = pf.shape[0]
        out = torch.empty_like(pf)
        save = torch.empty(n, 21, dtype=torch.float64, device=pf.device) if do_svd else None
        L.check(L.load().nefes_regressed_pose_fwd(n, _chk(pf, "pose"), 1 if do_svd else 0, float(t_scale), float(move[0]), float(move[1]), float(move[2]),
                                                  _chk(out, "out"), _chk(save, "save", torch.float64), _stream()), "nefes_regressed_pose_fwd")
        if do_svd:
            ctx.save_for_backward(save)
        ctx.shape, ctx.do_svd, ctx.t_scale = pose.shape, bool(do_svd), float(t_scale)
        return out.reshape(pose.shape)

    @staticmethod
    def backward(ctx, g):
        save = ctx.saved_tensors[0] if ctx.do_svd else None
        gf = _f32(g).reshape(-1, 3, 4)
        g_pose = torch.empty_like(gf)
        L.check(L.load().nefes_regressed_pose_bwd(gf.shape[0], _chk(save, "save", torch.float64), 1 if ctx.do_svd else 0, ctx.t_scale,
                                                  _chk(gf, "g_out"), _chk(g_pose, "g_pose"), _stream()), "nefes_regressed_pose_bwd")
        return g_pose.reshape(ctx.shape), None, None, None


def regressed_pose(pose, do_svd, world_setup=None):
    """[..., 3, 4] regressed poses -> svd_reg (if do_svd) -> fix_coord_supp(world_setup) (identity translation map when None)."""
    if world_setup is None:
        sc, mv = 1.0, (0., 0., 0.)
    else:
        s1, s2 = float(world_setup["pose_scale"]), float(world_setup["pose_scale2"])
        sc, mv = s1 * s2, tuple(float(v) * s2 for v in world_setup["move_all_cam_vec"])
    return RegressedPose.apply(pose, bool(do_svd), sc, mv)


# Where the next feature-loss Function writes its value: PoseRefiner points this at its static loss buffer around its loss call, so that an
# iteration needs no separate 4-byte copy kernel to leave the value there (None: a fresh tensor, as any other caller gets).
LOSS_OUT = None


def _loss_out(device):
    t = LOSS_OUT
    if t is not None and t.dim() == 0 and t.dtype == torch.float32 and t.device == torch.device(device) and not t.requires_grad:
        return t.detach()          # (a new tensor object on the same storage: autograd attaches this call's history to it, not to the buffer)
    return torch.empty((), device=device)


class CosineFeatureLoss(torch.autograd.Function):
    """feature_loss (dm/DFM_pose_refine.py:211-233, per_pixel=False) on [C, ...] maps: 1 - mean over channels of the cosine
    similarity over pixels; two launches forward, one backward (to `a` only: the target carries no gradient)."""

    @staticmethod
    def forward(ctx, a, b):
        Cc = a.shape[0]
        af, bf = _f32(a).reshape(Cc, -1), _f32(b).reshape(Cc, -1)
        if af.shape != bf.shape:
            raise ValueError(f"nefes_amd: feature maps of different shapes: {tuple(a.shape)} vs {tuple(b.shape)}")
        lib = L.load()
        scratch = torch.empty(lib.nefes_cosine_loss_scratch_doubles(Cc), dtype=torch.float64, device=af.device)
        loss = _loss_out(af.device)
        L.check(lib.nefes_cosine_loss_fwd(Cc, af.shape[1], _chk(af, "a"), _chk(bf, "b"), _chk(scratch, "scratch", torch.float64),
                                          _chk(loss, "loss"), _stream()), "nefes_cosine_loss_fwd")
        ctx.save_for_backward(af, bf, scratch)
        ctx.shape = a.shape
        cos = scratch[:4 * Cc].view(Cc, 4)[:, 3]          # float64 cosine similarity per channel (a view of the scratch)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for `cos` in the backward
        ctx.mark_non_differentiable(cos)
        return loss, cos

    @staticmethod
    def backward(ctx, g, _g_cos):
        if g is None:
            return (None,) * 2
        af, bf, scratch = ctx.saved_tensors
        gf = _f32(g).reshape(1)
        g_a = torch.empty_like(af)
        L.check(L.load().nefes_cosine_loss_bwd(af.shape[0], af.shape[1], _chk(af, "a"), _chk(bf, "b"),
                                               _chk(scratch, "scratch", torch.float64), _chk(gf, "g_loss"), _chk(g_a, "g_a"), _stream()),
                "nefes_cosine_loss_bwd")
        return g_a.reshape(ctx.shape), None


def cosine_feature_loss(a, b, return_cos=False):
    """1 - mean over the leading (channel) dimension of the cosine similarity over everything else; `return_cos` adds the
    per-channel similarities (float64, no gradient) -- per-image losses of a batch folded into the channel dimension."""
    loss, cos = CosineFeatureLoss.apply(a, b)
    return (loss, cos) if return_cos else loss


def psnr_ssim(x, y):
    """The verification step's two numbers (DFM_APR_refine.py:117-128, :146-150) of two [C,H,W] fp32 images on the GPU:
    (mse2psnr(img2mse(x, y)), SSIM()(x, y).mean()) as 0-d tensors -- two launches (csrc/refine.hip psnr_ssim_*) where the torch
    expressions take ~25.  Rows must be dense (stride 1 along W); row and plane strides are free, so a cropped view is not copied."""
    for name, t in (("x", x), ("y", y)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1):
            raise RuntimeError(f"nefes_amd: psnr_ssim: `{name}` must be a [C,H,W] float32 GPU tensor with dense rows (got {tuple(t.shape)}, "
                               f"strides {t.stride()}, {t.dtype}, {t.device})")
    if x.shape != y.shape:
        raise RuntimeError(f"nefes_amd: psnr_ssim: shapes differ ({tuple(x.shape)} vs {tuple(y.shape)})")
    lib = L.load()
    Cc, H, W = (int(v) for v in x.shape)
    ws = torch.empty(lib.nefes_psnr_ssim_workspace(Cc, H, W), dtype=torch.uint8, device=x.device)
    out = torch.empty(2, device=x.device)
    x, y = x.detach(), y.detach()
    L.check(lib.nefes_psnr_ssim(Cc, H, W, C.c_void_p(x.data_ptr()), x.stride(1), x.stride(0), C.c_void_p(y.data_ptr()), y.stride(1),
                                y.stride(0), _chk(ws, "workspace", torch.uint8), _chk(out, "out"), _stream()), "nefes_psnr_ssim")
    return out[0], out[1]


_GATHER_TABLES = {}


def _gather_T(n_in, n_out):
    return 4 * ((n_out + n_in - 1) // n_in) + 8


def bicubic_gather_table(n_in, n_out, o0, n_win, device, T=None):
    """(first, count, wt, T) of one axis (include/nefes_hip.h nefes_bicubic_gather_table), built once per geometry and device.  T: row
    length of `wt` (default: what this axis needs; the kernels that take the tables of BOTH axes take one T: `bicubic_gather_tables`)."""
    T = _gather_T(n_in, n_out) if T is None else int(T)
    key = (int(n_in), int(n_out), int(o0), int(n_win), T, str(device))
    t = _GATHER_TABLES.get(key)
    if t is None:
        first = torch.zeros(n_in, dtype=torch.int32, device=device)
        count = torch.zeros(n_in, dtype=torch.int32, device=device)
        wt = torch.zeros(n_in, T, device=device)
        L.check(L.load().nefes_bicubic_gather_table(n_in, n_out, o0, n_win, T, first.data_ptr(), count.data_ptr(), wt.data_ptr(), _stream()),
                "nefes_bicubic_gather_table")
        t = _GATHER_TABLES[key] = (first, count, wt, T)
    return t


def bicubic_gather_tables(h, w, OH, OW, crop, device):
    """The x and y tables with ONE row length (nefes_upcos_loss_bwd / nefes_upcos_prepare take a single T): until round 5 the y table was
    built with its own T and read with the x table's -- the same number for the loop's integer scale, not for OH / h != OW / w."""
    T = max(_gather_T(w, OW), _gather_T(h, OH))
    return (bicubic_gather_table(w, OW, crop, OW - 2 * crop, device, T), bicubic_gather_table(h, OH, crop, OH - 2 * crop, device, T))


class UpsampledCosineLoss(torch.autograd.Function):
    """cosine_feature_loss(bicubic_upsample(x, size, crop), target) for x [C,h,w] (or [1,C,h,w]) without the up-sampled image:
    csrc/refine.hip upcos kernels.  Returns (loss, per-channel cosine similarities) like CosineFeatureLoss."""

    @staticmethod
    def forward(ctx, x, target, OH, OW, crop):
        xf = _f32(x)
        Cc, h, w = xf.shape[-3:]
        xf = xf.reshape(Cc, h, w)
        tf = _f32(target).reshape(Cc, OH - 2 * crop, OW - 2 * crop)
        lib = L.load()
        scratch = torch.empty(lib.nefes_cosine_loss_scratch_doubles(Cc), dtype=torch.float64, device=xf.device)
        loss = _loss_out(xf.device)
        with _timed("upcos_loss_fwd"):
            L.check(lib.nefes_upcos_loss_fwd(Cc, h, w, OH, OW, crop, _chk(xf, "x"), _chk(tf, "target"), _chk(scratch, "scratch", torch.float64),
                                             _chk(loss, "loss"), _stream()), "nefes_upcos_loss_fwd")
        ctx.save_for_backward(xf, tf, scratch)
        ctx.cfg = (x.shape, OH, OW, crop)
        cos = scratch[:4 * Cc].view(Cc, 4)[:, 3]
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for `cos` in the backward
        ctx.mark_non_differentiable(cos)
        return loss, cos

    @staticmethod
    def backward(ctx, g, _g_cos):
        if g is None:
            return (None,) * 5
        xf, tf, scratch = ctx.saved_tensors
        shape, OH, OW, crop = ctx.cfg
        Cc, h, w = xf.shape
        gf = _f32(g).reshape(1)
        tmp = torch.empty(Cc, OH - 2 * crop, w, device=xf.device)
        g_x = torch.empty_like(xf)
        (fx, cx, wx, T), (fy, cy, wy, _) = bicubic_gather_tables(h, w, OH, OW, crop, xf.device)
        with _timed("upcos_loss_bwd"):
            L.check(L.load().nefes_upcos_loss_bwd(Cc, h, w, OH, OW, crop, _chk(xf, "x"), _chk(tf, "target"), _chk(scratch, "scratch", torch.float64),
                                                  _chk(gf, "g_loss"), fx.data_ptr(), cx.data_ptr(), wx.data_ptr(), fy.data_ptr(), cy.data_ptr(),
                                                  wy.data_ptr(), T, _chk(tmp, "tmp"), _chk(g_x, "g_x"), _stream()), "nefes_upcos_loss_bwd")
        return g_x.reshape(shape), None, None, None, None


def upsampled_cosine_loss(x, target, size, crop=0, return_cos=False):
    """1 - mean over channels of the cosine similarity (over pixels) between the bicubically up-sampled, cropped x and target."""
    loss, cos = UpsampledCosineLoss.apply(x, target, int(size[0]), int(size[1]), int(crop))
    return (loss, cos) if return_cos else loss


_GRAMS = {}


def bicubic_gram(n_in, n_out, o0, n_win, device):
    """(G [n_in, n_in] float64, half band width) of one axis' window (include/nefes_hip.h nefes_bicubic_gram), once per geometry."""
    key = (int(n_in), int(n_out), int(o0), int(n_win), str(device))
    t = _GRAMS.get(key)
    if t is None:
        G = torch.zeros(n_in, n_in, dtype=torch.float64, device=device)
        L.check(L.load().nefes_bicubic_gram(n_in, n_out, o0, n_win, G.data_ptr(), _stream()), "nefes_bicubic_gram")
        nz = G.nonzero()
        band = int((nz[:, 0] - nz[:, 1]).abs().max()) if nz.numel() else 0
        t = _GRAMS[key] = (G, band)
    return t


UPCOS_GRAM_LDS_MAX = 96 * 1024          # csrc/refine.hip NEFES_UPCOS_GRAM_LDS_MAX


class UpcosTarget:
    """What upsampled_cosine_loss needs of a FIXED target [C, OH-2crop, OW-2crop] (the refinement loop's: one target for all
    iterations of an image): Tt = Uy^T target Ux [C,h,w], |target|^2 per channel, and the Gram matrices of the two axes.  `update(target)`
    refills the same buffers (a captured graph keeps reading them).  csrc/refine.hip upcos_prep_* / upcos_gram_*."""

    def __init__(self, C, h, w, OH, OW, crop, device):
        self.C, self.h, self.w, self.OH, self.OW, self.crop = int(C), int(h), int(w), int(OH), int(OW), int(crop)
        CH, CW = self.OH - 2 * self.crop, self.OW - 2 * self.crop
        self.tt = torch.zeros(self.C, self.h, self.w, dtype=torch.float64, device=device)
        self.dbb = torch.zeros(self.C, dtype=torch.float64, device=device)
        self.gx, bx = bicubic_gram(self.w, self.OW, self.crop, CW, device)
        self.gy, by = bicubic_gram(self.h, self.OH, self.crop, CH, device)
        self.band = max(bx, by)
        self.tx, self.ty = bicubic_gather_tables(self.h, self.w, self.OH, self.OW, self.crop, device)
        self.device = device
        # one workgroup of the Gram-form forward holds a band of rows in LDS: geometries beyond its ceiling (h ~ 240 with w ~ 427, or
        # a wide band from a non-integer scale) stay on the one-pass kernels -- decided HERE, not by an error in the middle of a loop
        self.fits = int(L.load().nefes_upcos_gram_lds_bytes(self.h, self.w, self.band)) <= UPCOS_GRAM_LDS_MAX

    def update(self, target):
        CH, CW = self.OH - 2 * self.crop, self.OW - 2 * self.crop
        tf = _f32(target).reshape(self.C, CH, CW)
        tmp = torch.empty(self.C, CH, self.w, dtype=torch.float64, device=tf.device)
        (fx, cx, wx, T), (fy, cy, wy, _) = self.tx, self.ty
        L.check(L.load().nefes_upcos_prepare(self.C, self.h, self.w, self.OH, self.OW, self.crop, _chk(tf, "target"), fx.data_ptr(), cx.data_ptr(),
                                             wx.data_ptr(), fy.data_ptr(), cy.data_ptr(), wy.data_ptr(), T, _chk(tmp, "tmp", torch.float64),
                                             _chk(self.tt, "tt", torch.float64), _chk(self.dbb, "dbb", torch.float64), _stream()),
                "nefes_upcos_prepare")
        return self


class UpsampledCosineLossPrepared(torch.autograd.Function):
    """UpsampledCosineLoss against a prepared target (UpcosTarget): two small launches forward, one backward, on [C,h,w] data only."""

    @staticmethod
    def forward(ctx, x, prep):
        xf = _f32(x)
        if tuple(xf.shape[-3:]) != (prep.C, prep.h, prep.w) or xf.numel() != prep.C * prep.h * prep.w:
            raise ValueError(f"nefes_amd: features {tuple(x.shape)} do not match the prepared target ({prep.C}, {prep.h}, {prep.w})")
        xf = xf.reshape(prep.C, prep.h, prep.w)
        lib = L.load()
        scratch = torch.empty(lib.nefes_cosine_loss_scratch_doubles(prep.C), dtype=torch.float64, device=xf.device)
        pmat = torch.empty(prep.C, prep.h, prep.w, dtype=torch.float64, device=xf.device)
        loss = _loss_out(xf.device)
        with _timed("upcos_gram_fwd"):
            L.check(lib.nefes_upcos_gram_fwd(prep.C, prep.h, prep.w, _chk(xf, "x"), _chk(prep.tt, "tt", torch.float64),
                                             _chk(prep.dbb, "dbb", torch.float64), _chk(prep.gx, "gram_x", torch.float64),
                                             _chk(prep.gy, "gram_y", torch.float64), prep.band, _chk(scratch, "scratch", torch.float64),
                                             _chk(pmat, "pmat", torch.float64), _chk(loss, "loss"), _stream()), "nefes_upcos_gram_fwd")
        ctx.save_for_backward(scratch, pmat)
        ctx.prep, ctx.shape = prep, x.shape
        cos = scratch[:4 * prep.C].view(prep.C, 4)[:, 3]
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for `cos` in the backward
        ctx.mark_non_differentiable(cos)
        return loss, cos

    @staticmethod
    def backward(ctx, g, _g_cos):
        if g is None:
            return (None,) * 2
        scratch, pmat = ctx.saved_tensors
        prep = ctx.prep
        gf = _f32(g).reshape(1)
        g_x = torch.empty(prep.C, prep.h, prep.w, device=pmat.device)
        with _timed("upcos_gram_bwd"):
            L.check(L.load().nefes_upcos_gram_bwd(prep.C, prep.h, prep.w, _chk(prep.tt, "tt", torch.float64), _chk(pmat, "pmat", torch.float64),
                                                  _chk(scratch, "scratch", torch.float64), _chk(gf, "g_loss"), _chk(g_x, "g_x"), _stream()),
                    "nefes_upcos_gram_bwd")
        return g_x.reshape(ctx.shape), None


def upsampled_cosine_loss_prepared(x, prep, return_cos=False):
    """upsampled_cosine_loss(x, target, (OH, OW), crop) for the target `prep` (an UpcosTarget) was last updated with."""
    loss, cos = UpsampledCosineLossPrepared.apply(x, prep)
    return (loss, cos) if return_cos else loss


class FusionInput(torch.autograd.Function):
    """The rendered maps -> FusionNet's input in one launch (csrc/refine.hip fusion_input): affine colour transform with the image's
    12 exposure coefficients (or none), colour normalisation, [N,3] / [N,C] -> [B,3+C,H,W].  Gradients to rgb and feat; the
    coefficients are constants here (a trainable exposure network takes the torch path)."""

    @staticmethod
    def forward(ctx, rgb, feat, affine, B, H, W, mean, std):
        rgb, feat = _f32(rgb).reshape(-1, 3), _f32(feat)
        Cf = feat.shape[-1]
        feat = feat.reshape(-1, Cf)
        HW = int(H) * int(W)
        x = torch.empty(B, 3 + Cf, int(H), int(W), device=rgb.device)
        y = torch.empty_like(rgb) if affine is not None else None
        m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
        L.check(L.load().nefes_fusion_input_fwd(B, HW, Cf, _chk(rgb, "rgb"), _chk(feat, "feat"), _chk(affine, "affine"), m3, s3,
                                                _chk(x, "x"), _chk(y, "y"), _stream()), "nefes_fusion_input_fwd")
        ctx.save_for_backward(affine, y)
        ctx.cfg = (B, HW, Cf, s3)
        return x

    @staticmethod
    def backward(ctx, g_x):
        affine, y = ctx.saved_tensors
        B, HW, Cf, s3 = ctx.cfg
        g = _f32(g_x)
        g_rgb = torch.empty(B * HW, 3, device=g.device) if ctx.needs_input_grad[0] else None
        g_feat = torch.empty(B * HW, Cf, device=g.device) if ctx.needs_input_grad[1] else None
        L.check(L.load().nefes_fusion_input_bwd(B, HW, Cf, _chk(g, "g_x"), _chk(affine, "affine"), _chk(y, "y"), s3, _chk(g_rgb, "g_rgb"),
                                                _chk(g_feat, "g_feat"), _stream()), "nefes_fusion_input_bwd")
        return g_rgb, g_feat, None, None, None, None, None, None


def fusion_input(rgb, feat, affine, B, H, W, mean, std):
    return FusionInput.apply(rgb, feat, affine, int(B), int(H), int(W), tuple(mean), tuple(std))


class FusedAdam:
    """torch.optim.Adam (defaults: no weight decay, no amsgrad) over a few small fp32 parameters with a learning rate per
    parameter, as ONE launch per step (csrc/refine.hip adam_kernel): the parameters are views into one flat buffer, so are their
    .grad tensors, and the state (m, v, step) lives on the device -- capturable in a HIP graph by construction."""

    def __init__(self, params_and_lrs, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p, _ in params_and_lrs]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.step_t = torch.zeros(1, device=dev)
        self.lr = torch.cat([torch.full((p.numel(),), float(lr), dtype=torch.float64) for p, lr in params_and_lrs]).to(dev)
        self.betas, self.eps = betas, eps
        off = 0
        with torch.no_grad():
            for p in self.params:                     # re-home each parameter (and its gradient) as a view of the flat buffers
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                off += k

    def zero_state(self):
        self.m.zero_(); self.v.zero_(); self.step_t.zero_(); self.grad.zero_()

    def step(self):
        L.check(L.load().nefes_adam_step(self.flat.numel(), self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.step_t.data_ptr(), self.lr.data_ptr(), float(self.betas[0]), float(self.betas[1]),
                                         float(self.eps), _stream()), "nefes_adam_step")
