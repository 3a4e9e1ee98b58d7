"""CPU oracle for the multiresolution hash-grid encoding of BASELINE config 4.

TEST INFRASTRUCTURE ONLY (same rules as oracle/ref_cpu.py).

PARITY UNPINNED.  The reference only *configures* this encoding (script/models/nerfh_tcnn.py:60-75:
tcnn "HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
per_level_scale=exp(ln(2048/16)/15); input normalised by x01=(x+bound)/(2*bound), :151-156); the arithmetic lives
in NVlabs/tiny-cuda-nn, which the reference installs un-pinned from GitHub HEAD (README.md:24) and does not vendor,
and the reference holds no test or stored tensor at that boundary.  What follows restates tiny-cuda-nn's published
algorithm (Mueller et al. 2022, "Instant NGP", and tiny-cuda-nn's grid encoding):
    scale_l = base * b^l - 1,  res_l = ceil(scale_l) + 1,  pos = x01 * scale_l + 0.5,  cell = floor(pos), w = pos - cell
    entries_l = min(round_up(res_l^3, 8), 2^19)
    index = cell.x + cell.y*res_l + cell.z*res_l^2            if res_l^3 <= entries_l   (dense level)
          = (cell.x*1) ^ (cell.y*2654435761) ^ (cell.z*805459861)   otherwise (uint32 arithmetic)
    index %= entries_l ;  feature = trilinear blend of the 8 corner entries.
The HIP kernels are pinned against THIS file only; reports must say so.
"""
import math

import numpy as np
import torch

N_LEVELS, N_FEAT, LOG2_T, BASE_RES, MAX_RES = 16, 2, 19, 16, 2048
PER_LEVEL_SCALE = math.exp(math.log(MAX_RES / BASE_RES) / (N_LEVELS - 1))
PRIMES = (1, 2654435761, 805459861)


def level_geometry():
    """[(scale, res, entries, offset, hashed)] per level; offsets in entries."""
    out, off = [], 0
    for l in range(N_LEVELS):
        # per_level_scale travels through the C ABI as fp32; the level scale is evaluated in f64 and rounded once
        scale = np.float32(BASE_RES * float(np.float32(PER_LEVEL_SCALE)) ** l - 1.0)
        res = int(math.ceil(float(scale))) + 1
        entries = min((res ** 3 + 7) // 8 * 8, 1 << LOG2_T)
        out.append((float(scale), res, entries, off, res ** 3 > entries))
        off += entries
    return out, off


def table_entries():
    return level_geometry()[1]


def make_table(seed=0):
    """U(-1e-4, 1e-4) like tiny-cuda-nn's default grid initialisation (unverifiable from the reference)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(table_entries(), N_FEAT, generator=g) * 2 - 1) * 1e-4


def encode(x, table, bound):
    """x [M,3] in [-bound, bound] -> [M, 32].  Differentiable w.r.t. x (torch autograd through the blend weights)."""
    geo, _ = level_geometry()
    x01 = (x + bound) / (2 * bound)
    outs = []
    for scale, res, entries, off, hashed in geo:
        pos = x01 * scale + 0.5
        cell = torch.floor(pos).detach()
        w = pos - cell
        ci = cell.to(torch.int64)
        feat = 0
        for corner in range(8):
            d = [(corner >> k) & 1 for k in range(3)]
            cc = [ci[:, k] + d[k] for k in range(3)]
            if hashed:
                idx = ((cc[0] * PRIMES[0]) & 0xFFFFFFFF) ^ ((cc[1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((cc[2] * PRIMES[2]) & 0xFFFFFFFF)
            else:
                idx = cc[0] + cc[1] * res + cc[2] * res * res
            idx = idx % entries
            wc = 1
            for k in range(3):
                wc = wc * (w[:, k] if d[k] else (1 - w[:, k]))
            feat = feat + wc[:, None] * table[off + idx]
        outs.append(feat)
    return torch.cat(outs, 1)
