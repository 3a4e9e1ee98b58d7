#!/usr/bin/env python3
"""Accuracy of the bf16x6 split product planned for the trunk layers, emulated on the CPU.

x = h + m + l with h, m, l bf16 (exact: 24 = 3 x 8 mantissa bits); keep the six cross terms hh, hm, mh, hl, lh, mm
(relative size of what is dropped: 2^-24).  Compared with a plain fp32 GEMM of the same 256 -> 256 layer against float64.
Output on this container (torch 2.10 CPU):
    fp32                                 max err / max|y| = 4.670e-07   rms = 4.158e-08
    bf16x6 (fp32 acc)                    max err / max|y| = 2.518e-07   rms = 1.826e-08
    bf16x3                               max err / max|y| = 5.048e-06   rms = 8.452e-07
    bf16x6 exact-acc (truncation only)   max err / max|y| = 7.286e-09   rms = 1.059e-09
"""
import torch

torch.manual_seed(0)


def split3(x):
    h = x.to(torch.bfloat16).to(torch.float32)
    r = x - h
    m = r.to(torch.bfloat16).to(torch.float32)
    return h, m, (r - m).to(torch.bfloat16).to(torch.float32)


K, M, N = 256, 256, 4096
W = (torch.rand(M, K) * 2 - 1) / 16                      # nn.Linear init scale 1/sqrt(256)
X = torch.relu(torch.randn(K, N) * 0.3)
ref = W.double() @ X.double()
Wh, Wm, Wl = split3(W)
Xh, Xm, Xl = split3(X)
assert float((W - (Wh + Wm + Wl)).abs().max()) == 0. and float((X - (Xh + Xm + Xl)).abs().max()) == 0.
y6 = Wh @ Xh + Wh @ Xm + Wm @ Xh + Wh @ Xl + Wl @ Xh + Wm @ Xm
y3 = Wh @ Xh + Wh @ Xm + Wm @ Xh
d = lambda t: t.double()
y6d = d(Wh) @ d(Xh) + d(Wh) @ d(Xm) + d(Wm) @ d(Xh) + d(Wh) @ d(Xl) + d(Wl) @ d(Xh) + d(Wm) @ d(Xm)
sc = ref.abs().max()
for n, y in (('fp32', W @ X), ('bf16x6 (fp32 acc)', y6), ('bf16x3', y3), ('bf16x6 exact-acc (truncation only)', y6d)):
    print(f'{n:36s} max err / max|y| = {float((y.double() - ref).abs().max() / sc):.3e}   '
          f'rms = {float(((y.double() - ref) ** 2).mean().sqrt() / sc):.3e}')
