"""CPU oracle for the NeFeS render-and-refine hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain-torch restatement of the
reference's algorithm for the path named in BASELINE.json.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product (`nefes_amd/`) never does and fails loudly when the HIP library is
missing.

Parity status: PINNED.  `tools/make_goldens.py` imports the real reference from
/root/reference in the build container and stores its inputs/outputs under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function below
against those vectors (bit-exact for the fp32 op sequence, and against the live
import when /root/reference is present).

Every function cites the reference lines it follows (paths are relative to
/root/reference/).  All functions are dtype-generic: run them on float32
tensors for the reference's fp32 behaviour, or on float64 tensors to get the
"ground truth" used to judge gradient noise (SURVEY.md §7 hard part 10).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# a1/a2: ray generation                                script/models/ray_utils.py
# --------------------------------------------------------------------------
def pixel_dirs(H: int, W: int, focal: float, dtype=torch.float32) -> Tensor:
    """Camera-frame direction per pixel, [H,W,3]  (ray_utils.py:7-10).

    No +0.5 pixel-centre offset; x grows with column, y shrinks with row, z=-1.
    """
    cols = torch.linspace(0, W - 1, W, dtype=dtype)
    rows = torch.linspace(0, H - 1, H, dtype=dtype)
    ii = cols[None, :].expand(H, W)
    jj = rows[:, None].expand(H, W)
    return torch.stack([(ii - W * .5) / focal, -(jj - H * .5) / focal, -torch.ones_like(ii)], -1)


def ray_bundle(H: int, W: int, focal: float, c2w: Tensor):
    """rays_o, rays_d [H,W,3] from a 3x4 camera-to-world pose (ray_utils.py:5-16)."""
    d_cam = pixel_dirs(H, W, focal, c2w.dtype)
    rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], -1)       # :13
    rays_o = c2w[:3, -1].expand(rays_d.shape)                       # :15
    return rays_o, rays_d


def ndc_warp(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor):
    """Forward-facing NDC warp (ray_utils.py:27-44)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# a6: frequency encoding                    script/models/nerfh_nff.py:234-270
# --------------------------------------------------------------------------
def freq_encode(x: Tensor, n_freqs: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], each term
    over the whole 3-vector (nerfh_nff.py:247-267; log-sampled bands :253)."""
    if n_freqs == 0:                                                    # reduce_embedding = 1: the inputs themselves (:264-268, :318-326)
        return x
    bands = 2. ** torch.linspace(0., n_freqs - 1, steps=n_freqs)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(x * f.to(x.dtype)))
        parts.append(torch.cos(x * f.to(x.dtype)))
    return torch.cat(parts, -1)


# --------------------------------------------------------------------------
# a8: the field MLP                          script/models/nerfh_nff.py:421-576
# --------------------------------------------------------------------------
def field_param_shapes(typ: str, Wd: int, C: int, in_xyz: int = 63, in_dir: int = 27, D: int = 8, skip: int = 4):
    """Ordered (name, out, in) list in the construction order of the reference
    ctor (nerfh_nff.py:452-505) so that seeding reproduces its initial values."""
    spec = []
    for i in range(D):
        k = in_xyz if i == 0 else (Wd + in_xyz if i == skip else Wd)
        spec.append((f"xyz_encoding_{i + 1}.0", Wd, k))
    spec.append(("xyz_encoding_final", Wd, Wd))
    spec.append(("dir_encoding.0", Wd // 2, Wd + in_dir))
    spec.append(("static_sigma.0", 1, Wd))
    spec.append(("static_rgb.0", 3 + C, Wd // 2))
    if typ == "fine":
        spec.append(("transient_encoding.0", Wd // 2, Wd + in_dir))
        spec.append(("transient_encoding.2", Wd // 2, Wd // 2))
        spec.append(("transient_encoding.4", Wd // 2, Wd // 2))
        spec.append(("transient_sigma.0", 1, Wd // 2))
        spec.append(("transient_rgb.0", 3, Wd // 2))
        spec.append(("transient_beta.0", 1, Wd // 2))
    return spec


def make_field_params(typ: str, Wd: int = 256, C: int = 16, seed: int = 0, dtype=torch.float32, in_xyz: int = 63,
                      in_dir: int = 27) -> Dict[str, Tensor]:
    """Random-init parameters with the reference's key names.  Like the ctor
    (nerfh_nff.py:446) this reseeds the global generator, then draws each
    nn.Linear in construction order (default kaiming-uniform init)."""
    torch.manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, n_out, n_in in field_param_shapes(typ, Wd, C, in_xyz=in_xyz, in_dir=in_dir):
        lin = torch.nn.Linear(n_in, n_out)
        out[name + ".weight"] = lin.weight.detach().to(dtype).clone()
        out[name + ".bias"] = lin.bias.detach().to(dtype).clone()
    return out


def field_forward(p: Dict[str, Tensor], x: Tensor, sigma_only: bool = False, output_transient: bool = True,
                  in_xyz: int = 63, in_dir: int = 27, D: int = 8, skip: int = 4, act=None) -> Tensor:
    """NeRFH_NFF.forward on already-embedded inputs (nerfh_nff.py:525-576).
    `act` (tests only): callable (layer tag, pre-activation) -> activation replacing torch.relu for the hidden layers
    ("L1".."L8", "DIR", "T0", "T1", "T2") -- used to record pre-activations and to evaluate the network on a GIVEN
    ReLU branch pattern (the one a HIP forward pass took), where the function is smooth and gradients are comparable
    to 1e-6 instead of being dominated by which side of a kink a rounding error lands on."""
    lin = lambda name, t: F.linear(t, p[name + ".weight"], p[name + ".bias"])
    if act is not None:
        return _field_forward_act(p, x, sigma_only, output_transient, in_xyz, in_dir, D, skip, act)
    if sigma_only:
        e_xyz = x
    else:
        e_xyz, e_dir = torch.split(x, [in_xyz, in_dir], dim=-1)
    h = e_xyz
    for i in range(D):
        if i == skip:
            h = torch.cat([e_xyz, h], 1)                                # :551-552
        h = torch.relu(lin(f"xyz_encoding_{i + 1}.0", h))
    sigma = F.softplus(lin("static_sigma.0", h))                        # :485,555
    if sigma_only:
        return sigma
    feat = lin("xyz_encoding_final", h)                                 # no activation :559
    head_in = torch.cat([feat, e_dir], 1)
    g = torch.relu(lin("dir_encoding.0", head_in))
    static = torch.cat([lin("static_rgb.0", g), sigma], 1)              # no activation when C>0 :487-490
    if not output_transient:
        return static
    t = torch.relu(lin("transient_encoding.0", head_in))
    t = torch.relu(lin("transient_encoding.2", t))
    t = torch.relu(lin("transient_encoding.4", t))
    t_sigma = F.softplus(lin("transient_sigma.0", t))
    t_rgb = torch.sigmoid(lin("transient_rgb.0", t))
    t_beta = F.softplus(lin("transient_beta.0", t))
    return torch.cat([static, t_rgb, t_sigma, t_beta], 1)               # :563,573-576


def _field_forward_act(p, x, sigma_only, output_transient, in_xyz, in_dir, D, skip, act):
    lin = lambda name, t: F.linear(t, p[name + ".weight"], p[name + ".bias"])
    if sigma_only:
        e_xyz = x
    else:
        e_xyz, e_dir = torch.split(x, [in_xyz, in_dir], dim=-1)
    h = e_xyz
    for i in range(D):
        if i == skip:
            h = torch.cat([e_xyz, h], 1)
        h = act(f"L{i + 1}", lin(f"xyz_encoding_{i + 1}.0", h))
    sigma = F.softplus(lin("static_sigma.0", h))
    if sigma_only:
        return sigma
    feat = lin("xyz_encoding_final", h)
    head_in = torch.cat([feat, e_dir], 1)
    g = act("DIR", lin("dir_encoding.0", head_in))
    static = torch.cat([lin("static_rgb.0", g), sigma], 1)
    if not output_transient:
        return static
    t = act("T0", lin("transient_encoding.0", head_in))
    t = act("T1", lin("transient_encoding.2", t))
    t = act("T2", lin("transient_encoding.4", t))
    return torch.cat([static, torch.sigmoid(lin("transient_rgb.0", t)), F.softplus(lin("transient_sigma.0", t)),
                      F.softplus(lin("transient_beta.0", t))], 1)


# --------------------------------------------------------------------------
# a7: per-sample query with netchunk slicing   script/models/nerfh_nff.py:168-231
# --------------------------------------------------------------------------
def query_field(p: Dict[str, Tensor], pts: Tensor, viewdirs: Optional[Tensor], typ: str, output_transient: bool,
                test_time: bool, netchunk: int = 1 << 21, n_freq_xyz: int = 10, n_freq_dir: int = 4, act=None) -> Tensor:
    """run_network_NeRFH_NFF: [N,S,3] points (+ [N,3] view dirs) -> raw [N,S,R].
    `act`: see field_forward (tests only; called as act(tag, preact, row0) with the first flat sample index of the slice)."""
    flat = pts.reshape(-1, 3)
    sigma_only = (typ == "coarse" and test_time)                        # :192-202
    if not sigma_only:
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)       # :206-207,220-221
    outs = []
    for i in range(0, flat.shape[0], netchunk):
        e = freq_encode(flat[i:i + netchunk], n_freq_xyz)
        a = None if act is None else (lambda tag, pre, i=i: act(tag, pre, i))
        if sigma_only:
            outs.append(field_forward(p, e, sigma_only=True, in_xyz=e.shape[1], act=a))
        else:
            ed = freq_encode(dirs[i:i + netchunk], n_freq_dir)
            outs.append(field_forward(p, torch.cat([e, ed], 1), output_transient=output_transient, in_xyz=e.shape[1], in_dir=ed.shape[1],
                                      act=a))
    out = torch.cat(outs, 0)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


# --------------------------------------------------------------------------
# a9: alpha compositing                        script/models/nerfh_nff.py:25-166
# --------------------------------------------------------------------------
@dataclass
class Composite:
    rgb: Optional[Tensor]
    feat: Optional[Tensor]
    disp: Optional[Tensor]
    acc: Tensor
    weights: Tensor
    depth: Optional[Tensor]
    transient_sigmas: Optional[Tensor]
    beta: Optional[Tensor]


def composite(raw: Tensor, z: Tensor, raw_noise_std: float = 0., output_transient: bool = False, beta_min: float = 0.1,
              white_bkgd: bool = False, test_time: bool = False, typ: str = "coarse", store_rgb: bool = False,
              transient_at_test: bool = False) -> Composite:
    """raw2outputs_NeRFH_NFF with its four variants (A,B,C,D)."""
    sigma_only = (typ == "coarse" and test_time and not store_rgb)      # variant D :33-35
    t_sigma = None
    if sigma_only:
        s_sigma = raw[..., 0]
    else:
        n_col = raw.shape[-1] - (6 if output_transient else 1)          # :37-44
        s_col = raw[..., :n_col]
        s_sigma = raw[..., n_col]
        if output_transient:
            t_col = raw[..., n_col + 1:n_col + 4]
            t_sigma = raw[..., n_col + 4]
            t_beta = raw[..., n_col + 5]
    delta = z[:, 1:] - z[:, :-1]
    delta = torch.cat([delta, 1e2 * torch.ones_like(delta[:, :1])], -1)     # :55-60 (no |d| scaling)
    if output_transient:
        a_s = 1 - torch.exp(-delta * s_sigma)
        a_t = 1 - torch.exp(-delta * t_sigma)
        a = 1 - torch.exp(-delta * (s_sigma + t_sigma))
    else:
        noise = torch.randn_like(s_sigma) * raw_noise_std               # RNG advances even at std 0 (:67)
        a = 1 - torch.exp(-delta * (s_sigma + noise))
    T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a], -1)[:, :-1], -1)   # :71-72
    if output_transient:
        w_s = a_s * T
        w_t = a_t * T
    w = a * T
    acc = w.sum(-1)                                                     # einops reduce 'sum' :80-81
    if sigma_only:
        return Composite(None, None, None, acc, w, None, t_sigma, None)
    if output_transient:
        if test_time and not transient_at_test:                         # variant B :92-117
            T_s = torch.cumprod(torch.cat([torch.ones_like(a_s[:, :1]), 1 - a_s], -1)[:, :-1], -1)
            w_only = a_s * T_s
            rgb = (w_only[..., None] * s_col[..., :3]).sum(1)
            feat = (w_only.detach()[..., None] * s_col[..., 3:]).sum(1)
            depth = (w_only * z).sum(-1)
            disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / torch.sum(w_only, -1))
            beta = torch.zeros_like(acc)
            return Composite(rgb, feat, disp, acc, w_only, depth, t_sigma, beta)
        rgb_s = (w_s[..., None] * s_col[..., :3]).sum(1)                 # variant A :119-150
        feat = (w_s.detach()[..., None] * s_col[..., 3:]).sum(1)         # detached weights :122-125
        if white_bkgd:
            rgb_s = rgb_s + (1 - acc[:, None])
        rgb_t = (w_t[..., None] * t_col).sum(1)
        beta = (w_t * t_beta).sum(-1) + beta_min
        rgb = rgb_s + rgb_t
    else:                                                               # variant C :152-161
        rgb = (w[..., None] * s_col[..., :3]).sum(1)
        feat = (w.detach()[..., None] * s_col[..., 3:]).sum(1)
        beta = torch.zeros_like(acc)
    depth = torch.sum(w * z, -1)                                        # :163-166
    disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / torch.sum(w, -1))
    return Composite(rgb, feat, disp, acc, w, depth, t_sigma, beta)


# --------------------------------------------------------------------------
# a10: hierarchical sampling                  script/models/rendering.py:23-66
# --------------------------------------------------------------------------
def pdf_to_cdf(weights: Tensor) -> Tensor:
    """cdf [n, len(weights)+1] exactly as sample_pdf builds it (rendering.py:26-29)."""
    ww = weights + 1e-5
    pdf = ww / torch.sum(ww, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)


def invert_cdf(bins: Tensor, cdf: Tensor, u: Tensor):
    """searchsorted + linear interpolation (rendering.py:49-64).  Returns (samples, inds)."""
    u = u.contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c_lo = torch.gather(cdf, 1, below)
    c_hi = torch.gather(cdf, 1, above)
    b_lo = torch.gather(bins, 1, below)
    b_hi = torch.gather(bins, 1, above)
    denom = c_hi - c_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c_lo) / denom
    return b_lo + t * (b_hi - b_lo), inds


def inverse_cdf_samples(bins: Tensor, weights: Tensor, n: int, det: bool, u: Optional[Tensor] = None,
                        return_debug: bool = False):
    """sample_pdf (rendering.py:23-66).  `u` overrides the random draw when det=False."""
    cdf = pdf_to_cdf(weights)
    if det:
        u = torch.linspace(0., 1., steps=n, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n])
    elif u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [n], dtype=cdf.dtype)
    samples, inds = invert_cdf(bins, cdf, u)
    if return_debug:
        return samples, cdf, inds
    return samples


# --------------------------------------------------------------------------
# a5, a11-a13: per-chunk ray renderer        script/models/rendering.py:68-180
# --------------------------------------------------------------------------
@dataclass
class RenderCfg:
    """The subset of render_kwargs / args the path reads (nerfh_nff.py:710-735)."""
    N_samples: int = 64
    N_importance: int = 128
    perturb: float = 0.
    lindisp: bool = False
    white_bkgd: bool = False
    raw_noise_std: float = 0.
    test_time: bool = True
    transient_at_test: bool = True
    NeRFW: bool = True
    use_fine_only: bool = False
    netchunk: int = 1 << 21
    # octaves of the two embeddings (get_embedder, nerfh_nff.py:303-354): multires / multires_views by default; half of them with
    # reduce_embedding = 0 (:307-316: num_freqs = multires // 2 up to 2^((multires - 1) // 2)), none with reduce_embedding = 1
    n_freq_xyz: int = 10
    n_freq_dir: int = 4


def coarse_depths(near: Tensor, far: Tensor, n: int, lindisp: bool, t_rand: Optional[Tensor] = None) -> Tensor:
    """z_vals [n_rays, n] (rendering.py:96-112); `t_rand` = the stratified jitter draw."""
    t = torch.linspace(0., 1., steps=n, dtype=near.dtype)
    z = near * (1. - t) + far * t if not lindisp else 1. / (1. / near * (1. - t) + 1. / far * t)
    z = z.expand([near.shape[0], n])
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def render_rays(ray_batch: Tensor, p_coarse, p_fine, cfg: RenderCfg, t_rand: Optional[Tensor] = None,
                u_rand: Optional[Tensor] = None, debug: Optional[dict] = None, fine_act=None, z_fine=None,
                coarse_act=None) -> Dict[str, Tensor]:
    """rendering.py:68-180 for the nerfh_nff configuration.
    Tests only: `fine_act` = activation hook of the fine network (field_forward); `z_fine` = use these merged depths
    instead of the ones sampled here (to evaluate the fine pass at exactly the samples another implementation drew)."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, 8:11]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    if cfg.perturb > 0. and t_rand is None:
        t_rand = torch.rand(ray_batch.shape[0], cfg.N_samples, dtype=ray_batch.dtype)
    z = coarse_depths(near, far, cfg.N_samples, cfg.lindisp, t_rand if cfg.perturb > 0. else None)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]          # :114
    store_rgb = (cfg.N_importance == 0)
    raw = query_field(p_coarse, pts, viewdirs, "coarse", False, cfg.test_time, cfg.netchunk, cfg.n_freq_xyz, cfg.n_freq_dir,
                      act=coarse_act)                                                                            # :122
    c0 = composite(raw, z, cfg.raw_noise_std, white_bkgd=cfg.white_bkgd, test_time=cfg.test_time, typ="coarse",
                   store_rgb=store_rgb)
    out = c0
    z_samples = None
    if cfg.N_importance > 0:
        z_mid = .5 * (z[..., 1:] + z[..., :-1])
        z_samples, cdf, inds = inverse_cdf_samples(z_mid, c0.weights[..., 1:-1], cfg.N_importance,
                                                   det=(cfg.perturb == 0.), u=u_rand, return_debug=True)
        z_samples = z_samples.detach()                                            # :136
        z_coarse = z
        z = z_samples if cfg.use_fine_only else torch.sort(torch.cat([z, z_samples], -1), -1)[0]
        if z_fine is not None:
            z = z_fine.to(z.dtype)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]       # :142
        raw = query_field(p_fine, pts, viewdirs, "fine", cfg.NeRFW, cfg.test_time, cfg.netchunk, cfg.n_freq_xyz, cfg.n_freq_dir,
                          act=fine_act)
        out = composite(raw, z, cfg.raw_noise_std, output_transient=cfg.NeRFW, beta_min=0.1, white_bkgd=cfg.white_bkgd,
                        test_time=cfg.test_time, typ="fine", transient_at_test=cfg.transient_at_test)
        if debug is not None:
            debug.update(z_coarse=z_coarse, w_coarse=c0.weights, cdf=cdf, inds=inds, z_samples=z_samples, z_fine=z,
                         raw_fine=raw)
    ret = {"rgb_map": out.rgb, "disp_map": out.disp, "acc_map": out.acc, "feat_map": out.feat}
    if cfg.N_importance > 0 and not cfg.test_time:                                 # :160-173
        ret["rgb0"], ret["disp0"], ret["acc0"] = c0.rgb, c0.disp, c0.acc
        ret["z_std"] = torch.std(z_samples, dim=-1, unbiased=False)
        if cfg.NeRFW:
            ret["transient_sigmas"] = out.transient_sigmas
            ret["beta"] = out.beta
        if c0.feat is not None:
            ret["feat0"] = c0.feat
    return ret


def render(H: int, W: int, focal: float, p_coarse, p_fine, cfg: RenderCfg, chunk: int = 1024 * 32,
           rays=None, c2w: Optional[Tensor] = None, ndc: bool = False, near: float = 0., far: float = 1.,
           hist: Optional[Tensor] = None, debug: Optional[dict] = None, fine_act=None, z_fine=None, coarse_act=None):
    """rendering.py:197-243 (use_viewdirs=True).  Returns [rgb, disp, acc, extras].
    `fine_act` / `z_fine` (tests only, whole-batch; need chunk >= the ray count): see render_rays."""
    if c2w is not None:
        rays_o, rays_d = ray_bundle(H, W, focal, c2w)
    else:
        rays_o, rays_d = rays
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)                   # :217
    viewdirs = viewdirs.reshape(-1, 3)
    if ndc:
        rays_o, rays_d = ndc_warp(H, W, focal, 1., rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    ones = torch.ones_like(rays_d[..., :1])
    cols = [rays_o, rays_d, near * ones, far * ones, viewdirs]
    if hist is not None:
        cols.append(hist if hist.shape[0] == rays_o.shape[0] else hist.repeat(rays_o.shape[0], 1))
    bundle = torch.cat(cols, -1)
    pieces: Dict[str, list] = {}
    for i in range(0, bundle.shape[0], chunk):                                     # batchify_rays :182-195
        dbg = {} if debug is not None else None
        r = render_rays(bundle[i:i + chunk], p_coarse, p_fine, cfg, debug=dbg, fine_act=fine_act,
                        z_fine=None if z_fine is None else z_fine[i:i + chunk], coarse_act=coarse_act)
        if dbg:
            for k, v in dbg.items():
                debug.setdefault(k, []).append(v)
        for k, v in r.items():
            if v is not None:
                pieces.setdefault(k, []).append(v)
    full = {k: torch.cat(v, 0) for k, v in pieces.items()}
    if debug is not None:
        for k in list(debug.keys()):
            debug[k] = torch.cat(debug[k], 0)
    head = [full.get("rgb_map"), full.get("disp_map"), full.get("acc_map")]
    extras = {k: v for k, v in full.items() if k not in ("rgb_map", "disp_map", "acc_map")}
    return head + [extras]


# --------------------------------------------------------------------------
# measurement helpers
# --------------------------------------------------------------------------
def se3_exp_pose(r, t, dtype=torch.float32) -> Tensor:
    """[Exp(r) | t] via Rodrigues (script/utils/lie_group_helper.py:60-81 semantics)."""
    r = torch.tensor(r, dtype=torch.float64)
    th = torch.linalg.norm(r)
    K = torch.tensor([[0., -r[2], r[1]], [r[2], 0., -r[0]], [-r[1], r[0], 0.]], dtype=torch.float64)
    R = torch.eye(3, dtype=torch.float64) + (torch.sin(th) / th) * K + ((1 - torch.cos(th)) / th ** 2) * (K @ K)
    return torch.cat([R, torch.tensor(t, dtype=torch.float64)[:, None]], 1).to(dtype)


def bench_pose(dtype=torch.float32) -> Tensor:
    return se3_exp_pose((0.10, -0.20, 0.05), (0.10, 0.20, 0.30), dtype)


def bench_loss(rgb: Tensor, feat: Tensor) -> Tensor:
    """L = mean(feat^2) + mean(rgb^2)."""
    return (feat ** 2).mean() + (rgb ** 2).mean()
