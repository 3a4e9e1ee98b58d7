#!/usr/bin/env python3
"""The wait-state truth table of the GPU this runs on (csrc/hazard_probe.hip through nefes_probe_hazard): per producer ->
consumer pair the number of lanes x repetitions with a WRONG result at K = 0 ... 18 wait states.      python tools/hazard_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefes_amd import lib as L

TESTS = [("raw_f32_v   fp32 MFMA result -> vector read", 18), ("raw_f16_v   16-bit MFMA result -> v_mov", 12), ("raw_f16_a   ... -> v_accvgpr_read", 12),
         ("war_b       MFMA read of SrcB -> vector overwrite", None), ("war_c       MFMA read of SrcC -> vector overwrite", 7),
         ("valu_b      vector write of SrcB -> MFMA", 2), ("valu_c      vector write of SrcC -> MFMA", 2), ("vcc_valu    v_cmp VCC -> v_cndmask", 2),
         ("mfma_ab     MFMA result -> next MFMA SrcB", 12), ("waw_v       MFMA result -> vector overwrite", 12),
         ("raw_f16_lds MFMA result -> ds_write_b32", 12), ("valu_swap   vector write -> v_permlane32_swap", 2), ("trans_valu  v_exp -> vector read", 1),
         ("valu_dpp    vector write -> DPP read", 2), ("valu_readlane vector write -> v_readfirstlane", 1), ("accw_c      v_accvgpr_write SrcC -> MFMA", 2)]
KS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 18)


def table(blocks=1024, iters=200):
    lib = L.load()
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = {}
    for t, (name, _) in enumerate(TESTS):
        row = []
        for k in KS:
            cnt.zero_()
            L.check(lib.nefes_probe_hazard(t, k, blocks, iters, C.c_void_p(cnt.data_ptr()), None), "nefes_probe_hazard")
            torch.cuda.synchronize()
            row.append(int(cnt.item()) & 0xffffffff)
        out[name.split()[0]] = row
    return out


if __name__ == "__main__":
    tab = table()
    print(f"wrong lanes of {1024 * 256 * 200} at K = " + " ".join(str(k) for k in KS))
    for (name, llvm), row in zip(TESTS, tab.values()):
        print(f"{name:50s} LLVM {str(llvm):>4s} : " + " ".join(str(v) for v in row))
