#!/usr/bin/env python3
"""nefes_probe_store_hazard next to other work: a 16-byte store per lane whose third data register is overwritten `nops` + 1 wait
states later, alone on the device, next to a field forward (8 x 128 network) of another stream, and next to a plain copy kernel of
another stream.  Prints the number of stores that went out with the overwritten register (zeros in element 2) and which lanes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = L.load()
n = 4800 * 128 * 2
out = torch.empty(8, n, 4, device=dev)          # eight planes: every lane stores its four registers eight times back to back
fine = NeRFH_NFF('fine', W=128, f_dim=128, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk = fine.packed()
g = torch.Generator().manual_seed(1)
N, S = 4800, 128
ro = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(dev)
big = torch.empty(64 << 20, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def neighbour(kind):
    if kind == "field forward":
        with torch.no_grad():
            return ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)
    if kind == "copy kernel":
        return big.clone()
    return None


for kind in ("nothing", "copy kernel", "field forward"):
    for nops in (0, 1, 3, 7, 15):
        bad, lanes, total, planes, deltas = 0, set(), 0, set(), set()
        for rep in range(12):
            out.zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                keep = neighbour(kind)
            with torch.cuda.stream(s0):
                L.check(lib.nefes_probe_store_hazard(C.c_void_p(out.data_ptr()), n, nops, C.c_void_p(s0.cuda_stream)), "nefes_probe_store_hazard")
            torch.cuda.synchronize()
            want = ((torch.arange(n, device=dev) & 0xfffff) + 1).float()
            for e in (0, 1, 3):
                assert bool((out[:, :, e] == want[None]).all())
            want2 = want[None] + torch.arange(8, device=dev, dtype=torch.float32)[:, None]
            wrong = (out[:, :, 2] != want2).nonzero()
            if wrong.shape[0]:
                later = (out[:, :, 2] - want2)[wrong[:, 0], wrong[:, 1]]
                deltas |= set(later.unique().tolist())
            bad += wrong.shape[0]; total += 8 * n
            lanes |= set((wrong[:, 1] % 64).unique().tolist())
            planes |= set(wrong[:, 0].unique().tolist())
        print(f"next to {kind:14s} register re-used {nops + 1:2d} wait states behind each store: {bad:8d} of {total} stores went out with a LATER value"
              + (f"; lanes {sorted(lanes)[0]}..{sorted(lanes)[-1]}, stores {sorted(planes)} of the eight, value ahead by {sorted(deltas)}" if lanes else ""))


# ---- the packed multiply with operand selection (the instruction that produced composite_bwd4's wrong elements) ----
cnt = torch.zeros(4800 * 128 * 4, dtype=torch.int32, device=dev)
clk = (C.c_double(), C.c_double())


ga, gb = torch.randn(4096, 4096, device=dev, dtype=torch.float16), torch.randn(4096, 4096, device=dev, dtype=torch.float16)
fa, fb = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
fine256 = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk256 = fine256.packed()
coarse128 = NeRFH_NFF('coarse', W=128, f_dim=128).requires_grad_(False).to(dev)
pkc = coarse128.packed()


def neighbour2(kind):
    with torch.cuda.stream(s1), torch.no_grad():
        if kind == "fp16 GEMM (torch.mm)":
            return [ga @ gb for _ in range(6)]
        if kind == "fp32 GEMM (torch.mm)":
            return [fa @ fb for _ in range(2)]
        if kind.startswith("instruction loop"):
            k = int(kind.split("#")[1][0])
            L.check(lib.nefes_probe_aggressor(k, 60000, 512, C.c_void_p(sink.data_ptr()), C.c_void_p(s1.cuda_stream)), "nefes_probe_aggressor")
            return None
        if kind.startswith("field forward, split "):
            old = ops.SPLIT
            ops.SPLIT = kind.split("split ")[1]
            try:
                return ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)
            finally:
                ops.SPLIT = old
        if kind == "FusionNet convolution 5x5 (fp32 MFMA)":
            return [ops.frozen_conv2d(cx, cw, None) for _ in range(20)]
        if kind == "compositing forward":
            return [ops.composite_fwd(craw, z, 128, L.COMP_TRANSIENT, 0.03) for _ in range(4)]
        if kind == "field forward, width 256":
            return ops.FieldFromRays.apply(ro, rd, rd, z, pk256, L.FIELD_FULL)
        if kind == "sigma-only field forward":
            return ops.FieldFromRays.apply(ro, rd, rd, z, pkc, L.FIELD_SIGMA)
        return neighbour(kind)


sink = torch.zeros(2048 * 256, device=dev)
cx, cw = torch.randn(8, 64, 60, 80, device=dev), torch.randn(128, 64, 5, 5, device=dev)
with torch.no_grad():
    craw = ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)
torch.cuda.synchronize()                   # (the first measurement below is "next to nothing": nothing of the set-up may still be running)
KINDS = ("nothing", "copy kernel", "fp16 GEMM (torch.mm)", "fp32 GEMM (torch.mm)", "instruction loop #0 v_fma_mixlo/hi_f16", "instruction loop #1 MFMA 32x32x16 f16",
         "instruction loop #2 both", "instruction loop #3 v_pk_fma_f32 op_sel_hi", "instruction loop #4 v_fma_f32", "FusionNet convolution 5x5 (fp32 MFMA)",
         "compositing forward", "field forward, split f32", "field forward, split x6", "sigma-only field forward",
         "field forward, width 256", "field forward")
for kind in KINDS:
    tot = torch.zeros(4, dtype=torch.int64)
    lanes = set()
    for rep in range(12):
        keep = neighbour2(kind)
        with torch.cuda.stream(s0):
            L.check(lib.nefes_probe_pk_mul(C.c_void_p(cnt.data_ptr()), cnt.numel(), 64, C.c_void_p(s0.cuda_stream)), "nefes_probe_pk_mul")
        torch.cuda.synchronize()
        c = cnt.cpu()
        for b in range(4):
            tot[b] += int(((c >> (8 * b)) & 255).sum())
        lanes |= set((c.nonzero().flatten() % 64).unique().tolist())
    print(f"next to {kind:44s} wrong results of {12 * cnt.numel() * 64} each: v_pk_mul_f32 op_sel:[0,1] {int(tot[0])}, [1,0] {int(tot[1])}; v_pk_add_f32 op_sel:[0,1] {int(tot[2])}, "
          f"[1,0] {int(tot[3])}" + (f"; lanes {sorted(lanes)[0]}..{sorted(lanes)[-1]} ({len(lanes)} distinct)" if lanes else ""))
