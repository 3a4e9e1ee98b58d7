#!/usr/bin/env python3
"""Feasibility of a two-part fp16 split, emulated on the CPU through an 8-layer 256-wide ReLU MLP.

x = h + l with h = fp16(x * 2^e), l = fp16(x * 2^e - h): 11 + 11 = 22 significant bits, three products hh + hl + lh (half
the matrix-core work of bf16x6).  fp16 has a 5-bit exponent, so every operand is scaled by a power of two first: the
weights per layer (max |w| -> [2^13, 2^14)), the activations per SAMPLE and layer (max |x| of the column -> [2^13, 2^14));
powers of two commute with the products, ReLU and the sign bit the masks record, and are undone exactly at the end.
Compared against float64 next to plain fp32, bf16x6 and the three-product bf16 variant.
"""
import torch

torch.manual_seed(0)


def split_bf16(x, parts):
    out, r = [], x
    for _ in range(parts):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def split_fp16(x):
    h = x.to(torch.float16).to(torch.float32)
    return h, (x - h).to(torch.float16).to(torch.float32)


def pow2_scale(amax, target=13):
    """power of two s with amax * s in [2^target, 2^(target+1))"""
    e = torch.floor(torch.log2(amax.clamp_min(1e-30)))
    return torch.exp2(target - e)


def layer(W, X, kind):
    if kind == "f64":
        return W.double() @ X.double()
    if kind == "fp32":
        return W @ X
    if kind in ("bf16x6", "bf16x3"):
        Wp, Xp = split_bf16(W, 3), split_bf16(X, 3)
        y = Wp[0] @ Xp[0] + Wp[0] @ Xp[1] + Wp[1] @ Xp[0]
        if kind == "bf16x6":
            y = y + Wp[0] @ Xp[2] + Wp[2] @ Xp[0] + Wp[1] @ Xp[1]
        return y
    if kind == "fp16x3":
        sw = pow2_scale(W.abs().max())
        sx = pow2_scale(X.abs().amax(0, keepdim=True))          # per sample (column)
        Wh, Wl = split_fp16(W * sw)
        Xh, Xl = split_fp16(X * sx)
        return (Wh @ Xh + Wh @ Xl + Wl @ Xh) / (sw * sx)         # fp32 accumulation, exact power-of-two unscale
    raise ValueError(kind)


K, N, L = 256, 8192, 8
Ws = [(torch.rand(K, K) * 2 - 1) / 16 * 2.0 for _ in range(L)]      # gain ~ keeps the activations O(1) through the ReLUs
bs = [(torch.rand(K, 1) * 2 - 1) / 16 for _ in range(L)]
X0 = torch.randn(K, N) * 0.5
X0[:, : N // 4] *= 1e-3                                              # a quarter of the samples with tiny inputs
X0[:, N // 4: N // 2] *= 50.                                         # and a quarter with large ones
outs = {}
for kind in ("f64", "fp32", "bf16x6", "bf16x3", "fp16x3"):
    X = X0.double() if kind == "f64" else X0.clone()
    for W, b in zip(Ws, bs):
        Y = layer(W, X, kind) + (b.double() if kind == "f64" else b)
        X = torch.relu(Y)
    outs[kind] = Y.double()
ref = outs.pop("f64")
sc = ref.abs().amax(0, keepdim=True)                                 # per sample: small-input samples are judged on their own scale
for kind, y in outs.items():
    e = ((y - ref).abs() / sc)
    print(f"{kind:8s} after {L} layers: max err / max|y| per sample = {float(e.max()):.3e}   rms = {float((e ** 2).mean().sqrt()):.3e}")
