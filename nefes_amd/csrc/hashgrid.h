// Multiresolution hash-grid encoding: geometry and per-sample device functions shared by the stand-alone kernels (hashgrid.hip)
// and the field kernels that evaluate the encoding in their own prologue / epilogue (field_fwd_h3.hip / field_bwd_h3.hip,
// ENC = NEFES_XYZ_HASHGRID_FUSED).  Algorithm: oracle/hashgrid_ref.py (script/models/nerfh_tcnn.py:60-75,151-156 configures it;
// the arithmetic is tiny-cuda-nn's published one -- PARITY UNPINNED, see hashgrid.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nefes_hip.h"

#define HG_MAX_LEVELS 16
struct HgLevel {
    float scale;
    uint32_t res, entries, offset, hashed;
};
struct HgGeom {
    HgLevel lv[HG_MAX_LEVELS];
    int n_levels;
    float inv_range;   // 1 / (2 bound)
    float bound;
};

static inline int hg_geometry(const NefesHashGridDesc* d, HgGeom* g, uint64_t* total) {
    if (!d || d->n_levels <= 0 || d->n_levels > HG_MAX_LEVELS || d->n_features != 2 || d->log2_hashmap_size <= 0 ||
        d->log2_hashmap_size > 24 || !(d->bound > 0.f))
        return NEFES_E_UNSUPPORTED;
    uint64_t off = 0;
    g->n_levels = d->n_levels;
    g->bound = d->bound;
    g->inv_range = 1.f / (2.f * d->bound);
    for (int l = 0; l < d->n_levels; ++l) {
        // level scale evaluated in f64 from the fp32 growth factor and rounded once (oracle/hashgrid_ref.py does the same)
        const float scale = (float)((double)d->base_resolution * pow((double)d->per_level_scale, (double)l) - 1.0);
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint64_t dense = (uint64_t)res * res * res;
        uint64_t entries = (dense + 7) / 8 * 8;
        const uint64_t cap = 1ull << d->log2_hashmap_size;
        if (entries > cap) entries = cap;
        g->lv[l] = {scale, res, (uint32_t)entries, (uint32_t)off, dense > entries ? 1u : 0u};
        off += entries;
    }
    for (int l = d->n_levels; l < HG_MAX_LEVELS; ++l) g->lv[l] = {0.f, 1u, 8u, 0u, 0u};
    if (total) *total = off;
    return 0;
}

// index % entries without the division on the common path: a hashed level has entries = 2^log2_hashmap_size (mask); a dense
// level has entries >= res^3 and, for positions inside the bound, corner coordinates <= res, so its linear index is
// < 2 * entries (one conditional subtraction).  Positions outside the bound take the division (same result as before).
__device__ __forceinline__ uint32_t hg_index(const HgLevel& L, uint32_t x, uint32_t y, uint32_t z) {
    if (L.hashed) return L.offset + ((x ^ (y * 2654435761u) ^ (z * 805459861u)) & (L.entries - 1u));
    uint32_t i = x + y * L.res + z * L.res * L.res;
    if (i >= L.entries) {
        i -= L.entries;
        if (i >= L.entries) i %= L.entries;
    }
    return L.offset + i;
}

// cell and blend weights of position x (3 floats) at level L (nerfh_tcnn.py:156: x01 = (x + bound) / (2 bound))
__device__ __forceinline__ void hg_cell(const HgLevel& L, float bound, const float (&x)[3], uint32_t (&c)[3], float (&w)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x01 = (x[k] + bound) / (2.f * bound);
        const float pos = x01 * L.scale + 0.5f;
        const float fl = floorf(pos);
        c[k] = (uint32_t)(int)fl;
        w[k] = pos - fl;
    }
}
// the two features of level L at x: trilinear blend of the eight corner entries (corner order and fma chain as hashgrid_fwd_kernel)
__device__ __forceinline__ float2 hg_level_fwd(const HgLevel& L, float bound, const float2* __restrict__ table, const float (&x)[3]) {
    float w[3];
    uint32_t c[3];
    hg_cell(L, bound, x, c, w);
    float2 f[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
        f[corner] = table[hg_index(L, c[0] + (corner & 1), c[1] + ((corner >> 1) & 1), c[2] + (corner >> 2))];
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const float wc = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        acc.x = fmaf(wc, f[corner].x, acc.x);
        acc.y = fmaf(wc, f[corner].y, acc.y);
    }
    return acc;
}
// level L's contribution to d loss / d x, given the gradient `ge` of its two features (frozen table)
__device__ __forceinline__ void hg_level_bwd_x(const HgLevel& L, float bound, float inv_range, const float2* __restrict__ table,
                                               const float (&x)[3], float2 ge, float (&gx)[3]) {
    float w[3];
    uint32_t c[3];
    hg_cell(L, bound, x, c, w);
    float2 f[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
        f[corner] = table[hg_index(L, c[0] + (corner & 1), c[1] + ((corner >> 1) & 1), c[2] + (corner >> 2))];
    float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const float v = f[corner].x * ge.x + f[corner].y * ge.y;
        const float wx = dx ? w[0] : 1.f - w[0], wy = dy ? w[1] : 1.f - w[1], wz = dz ? w[2] : 1.f - w[2];
        g[0] += (dx ? v : -v) * wy * wz;
        g[1] += (dy ? v : -v) * wx * wz;
        g[2] += (dz ? v : -v) * wx * wy;
    }
    const float s = L.scale * inv_range;   // d pos / d x
#pragma unroll
    for (int k = 0; k < 3; ++k) gx[k] += g[k] * s;
}

// ---- the encoding evaluated INSIDE the field kernels (two lanes per sample: lane half h = 0 / 1) ---------------------------------
// The field kernels hold a sample's 32 features as sixteen "slots" per lane: slot s of lane half h = feature 2s + h = feature h of
// level s.  Lane half h evaluates BOTH features of the eight levels 2i + h (one 8-byte gather per corner instead of two 4-byte
// ones, half the index arithmetic per lane); one v_permlane32_swap per level pair then hands every lane its sixteen slots:
//     swap(P, Q) with P / Q = features 0 / 1 of "my" level:  first result  = [P of half 0, Q of half 0] = slot 2i   of halves 0 / 1,
//                                                           second result = [P of half 1, Q of half 1] = slot 2i+1 of halves 0 / 1.
// (the geometry `g` lives in LDS in the field kernels: level 2i + h is one per-lane read, no selects)
__device__ __forceinline__ HgLevel hg_pick_level(const HgGeom& g, int i, int h) { return g.lv[2 * i + h]; }
__device__ __forceinline__ void hg_encode_slots(float (&E)[16], const float (&x)[3], int h, const HgGeom& g, const float2* __restrict__ table) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float2 f = hg_level_fwd(hg_pick_level(g, i, h), g.bound, table, x);
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(f.x), __float_as_uint(f.y), false, false);
        E[2 * i] = __uint_as_float(r[0]);
        E[2 * i + 1] = __uint_as_float(r[1]);
    }
}
// gx += d loss / d x through the encoding.  This lane's slot gradients d loss / d feature (2s + h), s = 0..15, are read through
// `slot(s)` (the field backward parks them in LDS); both lane halves return their levels' share (the caller adds the two halves).
// Four levels per pass, two passes: all sixteen levels' gathers in flight at once need 128 registers at the one point of the kernel
// where hipcc then spills into the accumulator file and splits the accumulator tiles' live ranges (moves inside the asm-scheduled
// runs: tests/test_pack_stream.py).
template <class SlotFn>
__device__ __forceinline__ void hg_encode_slots_bwd(float (&gx)[3], const SlotFn& slot, const float (&x)[3], int h, const HgGeom& g,
                                                    const float2* __restrict__ table) {
#pragma unroll 1
    for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * ii + k;
            // inverse of the forward exchange: [slot 2i of halves 0 / 1] , [slot 2i+1 of halves 0 / 1]  ->  (d P, d Q) of my level
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(slot(2 * i)), __float_as_uint(slot(2 * i + 1)), false, false);
            hg_level_bwd_x(hg_pick_level(g, i, h), g.bound, g.inv_range, table, x, make_float2(__uint_as_float(r[0]), __uint_as_float(r[1])), gx);
        }
    }
}
