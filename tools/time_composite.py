#!/usr/bin/env python3
"""Kernel time and HBM-side rate of the compositing and sampling kernels at the benchmark shapes (algorithmic bytes of SURVEY.md
section 8d / DESIGN.md section 4.2-4.3).  NEFES_HIP_LIB selects a side build for A/B runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefes_amd import lib as L, ops

dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, N, S, C, Nc in (("G-metric 640x480, 64+128, C=16", 307200, 192, 16, 64), ("G-ref 80x60, 64+64, C=128", 4800, 128, 128, 64),
                          ("G-ref x 8 images", 38400, 128, 128, 64)):
    R = 3 + C + 6
    g = torch.Generator(device=dev).manual_seed(1)
    raw_t = torch.randn(N, R, S, device=dev, generator=g) * 0.5
    raw_t[:, 3 + C] = torch.nn.functional.softplus(raw_t[:, 3 + C] * 4)            # densities >= 0, as the field's heads emit them
    raw_t[:, 3 + C + 4] = torch.nn.functional.softplus(raw_t[:, 3 + C + 4] * 4 - 1)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4, -1)[0]
    flags = L.COMP_TRANSIENT
    outs = ops.composite_fwd(raw_t, z, C, flags)
    t_f = timed(lambda: ops.composite_fwd(raw_t, z, C, flags))
    g_rgb, g_feat = torch.randn(N, 3, device=dev), torch.randn(N, C, device=dev)
    g_raw = torch.empty_like(raw_t)
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    bwd = lambda: L.check(lib.nefes_composite_bwd(N, S, C, flags, raw_t.data_ptr(), z.data_ptr(), g_rgb.data_ptr(), g_feat.data_ptr(), None,
                                                  None, None, None, None, g_raw.data_ptr(), st), "bwd")
    t_b = timed(bwd)
    b_f = N * (S * R * 4 + S * 4 + (3 + C + 3) * 4)
    b_b = N * (2 * S * R * 4 + S * 4 + (3 + C) * 4)
    raw_c = raw_t[:, 3 + C:3 + C + 1, :Nc].contiguous()
    zc = z[:, :Nc].contiguous()
    t_d = timed(lambda: ops.composite_fwd(raw_c, zc, 0, L.COMP_SIGMA_ONLY))
    b_d = N * (Nc * 4 * 3 + 4)
    w = torch.rand(N, Nc, device=dev)
    t_s = timed(lambda: ops.sample_pdf_merge(zc, w, S - Nc))
    b_s = N * (Nc * 8 + S * 4)
    # the coarse pass behind its field kernel as one launch (csrc/sample_pdf.hip coarse_sample_kernel): sigma in, merged depths out,
    # on a depth row shared by every ray (what render() does at test time) -- realistic weights: a density bump along each ray
    sig = (4.0 * torch.exp(-0.5 * ((torch.arange(Nc, device=dev)[None] - torch.rand(N, 1, device=dev, generator=g) * Nc) / 3.0) ** 2)).contiguous()
    z_row = ops.coarse_depth_row(Nc, 0., 4., False, dev)
    t_c = timed(lambda: ops.coarse_sample(sig, z_row, S - Nc, want_samples=False))
    b_c = N * (Nc * 4 + S * 4)
    chk = [float(t.double().abs().sum()) for t in (outs[0], outs[1], outs[5], g_raw)]
    print("   checksums (|rgb|, |feat|, |weights|, |d raw|):", " ".join(f"{c:.9e}" for c in chk))
    print(f"{name}: composite_fwd {t_f:.3f} ms = {b_f / t_f / 1e9:.2f} TB/s | composite_bwd {t_b:.3f} ms = {b_b / t_b / 1e9:.2f} TB/s | "
          f"composite_fwd[D] {t_d:.3f} ms = {b_d / t_d / 1e9:.2f} TB/s | sample_pdf_merge {t_s:.3f} ms = {b_s / t_s / 1e9:.2f} TB/s | "
          f"coarse_sample (D + sample_pdf + merge, one launch) {t_c:.3f} ms = {b_c / t_c / 1e9:.2f} TB/s")
