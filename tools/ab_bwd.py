#!/usr/bin/env python3
"""A/B of two builds of the fp16 backward kernel on the same forward state: the shipped library against a side build
(tools/ab_bwd.py path/to/side.so [Wd C N S]); prints where the two d pts differ (per 128-sample tile, per wave, per lane)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF

side = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(side, name); fn.restype = res; fn.argtypes = args
Wd, Cf, N, S = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 16, 41, 24)
dev = "cuda"
net = NeRFH_NFF('fine', W=Wd, f_dim=Cf, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk = net.packed()
g = torch.Generator().manual_seed(3)
o = (torch.randn(N, 3, generator=g) * 0.3).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = torch.sort(torch.rand(N, S, generator=g) * 3.8 + 0.1, -1)[0].to(dev)
raw_t, masks = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=True)
g_raw = torch.randn(raw_t.shape, generator=torch.Generator().manual_seed(5)).to(dev)
outs = []
for lib in (L.load(), side):
    g_pts, g_vs = torch.empty(N * S, 3, device=dev), torch.empty(N * S, 3, device=dev)
    rc = lib.nefes_field_bwd_h3(pk.desc, pk.blob.data_ptr(), N, S, o.data_ptr(), d.data_ptr(), z.data_ptr(), None, d.data_ptr(),
                                raw_t.data_ptr(), g_raw.data_ptr(), masks.data_ptr(), g_pts.data_ptr(), None, g_vs.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    outs.append((g_pts.cpu().double(), g_vs.cpu().double()))
(a, av), (b, bv) = outs
sc = a.abs().max()
e = ((a - b).abs().max(1)[0] / sc)
print(f"max rel diff d pts {float(e.max()):.3e}   d viewdirs {float(((av - bv).abs().max() / av.abs().max())):.3e}   samples wrong (>1e-5): {int((e > 1e-5).sum())} of {e.numel()}")
M = N * S
for t in range((M + 127) // 128):
    seg = e[t * 128:(t + 1) * 128]
    waves = [float(seg[w * 32:(w + 1) * 32].max()) if seg.numel() > w * 32 else 0. for w in range(4)]
    bad = (seg > 1e-5).nonzero().flatten().tolist()
    print(f"tile {t}: per-wave max " + " ".join(f"{w:.1e}" for w in waves) + f"   bad lanes {bad[:40]}")
