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
