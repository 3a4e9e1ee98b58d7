// Multiresolution hash-grid positional encoding (BASELINE config 4) and its backward to x.
// The reference only configures this encoding (script/models/nerfh_tcnn.py:60-75, input normalisation :151-156);
// the arithmetic is tiny-cuda-nn's published algorithm, restated in oracle/hashgrid_ref.py (PARITY UNPINNED:
// tiny-cuda-nn is neither vendored nor version-pinned by the reference, and the model is orphaned there).
//
// Roofline: gather-bound.  Per sample: 16 levels x 8 corners x 8 B (fp32 x2) = 1 KiB of table reads, 12 B in, 128 B out.
// The 48.8 MB table stays in the 256 MiB Infinity Cache.  One thread per (sample, level): a wave covers 4 samples,
// writes 512 contiguous bytes, and the backward reduces the 16 levels of a sample with 4 xor-shuffles.
#include <hip/hip_runtime.h>
#include <math.h>

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

static int hg_geometry(const NefesHashGridDesc* d, HgGeom* g, uint64_t* total) {
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

__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(HgGeom g, const float2* __restrict__ table, long long M,
                                                           const float* __restrict__ x, float2* __restrict__ enc) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = tid / g.n_levels;
    const int l = (int)(tid - m * g.n_levels);
    if (m >= M) return;
    const HgLevel L = g.lv[l];
    float w[3];
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x01 = (x[m * 3 + k] + g.bound) / (2.f * g.bound);   // nerfh_tcnn.py:156
        const float pos = x01 * L.scale + 0.5f;
        const float fl = floorf(pos);
        c[k] = (uint32_t)(int)fl;
        w[k] = pos - fl;
    }
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const float wc = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        const float2 f = table[hg_index(L, c[0] + dx, c[1] + dy, c[2] + dz)];
        acc.x = fmaf(wc, f.x, acc.x);
        acc.y = fmaf(wc, f.y, acc.y);
    }
    // the embedding is written once and read once by the field kernel: a non-temporal store, so that 10 GB of output per fine pass
    // do not push the table (the data with reuse: 67 MB, four times the L2s) out of the caches on its way
    __builtin_nontemporal_store(acc.x, &enc[tid].x);     // enc[m][2l .. 2l+1]
    __builtin_nontemporal_store(acc.y, &enc[tid].y);
}

__global__ __launch_bounds__(256) void hashgrid_bwd_x_kernel(HgGeom g, const float2* __restrict__ table, long long M,
                                                             const float* __restrict__ x, const float2* __restrict__ g_enc,
                                                             float* __restrict__ g_x) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    long long m = tid / 16;                 // 16 lanes per sample (levels beyond n_levels contribute zero)
    const int l = (int)(tid & 15);
    const bool live = m < M && l < g.n_levels;
    if (m >= M) m = M - 1;
    float gx[3] = {0.f, 0.f, 0.f};
    if (live) {
        const HgLevel L = g.lv[l];
        float w[3];
        uint32_t c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x01 = (x[m * 3 + k] + g.bound) / (2.f * g.bound);
            const float pos = x01 * L.scale + 0.5f;
            const float fl = floorf(pos);
            c[k] = (uint32_t)(int)fl;
            w[k] = pos - fl;
        }
        const float* gep = (const float*)&g_enc[m * g.n_levels + l];                       // (read once: non-temporal)
        const float2 ge = make_float2(__builtin_nontemporal_load(gep), __builtin_nontemporal_load(gep + 1));
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const float2 f = table[hg_index(L, c[0] + dx, c[1] + dy, c[2] + dz)];
            const float v = f.x * ge.x + f.y * ge.y;
            const float wx = dx ? w[0] : 1.f - w[0], wy = dy ? w[1] : 1.f - w[1], wz = dz ? w[2] : 1.f - w[2];
            gx[0] += (dx ? v : -v) * wy * wz;
            gx[1] += (dy ? v : -v) * wx * wz;
            gx[2] += (dz ? v : -v) * wx * wy;
        }
        const float s = L.scale * g.inv_range;   // d pos / d x
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[k] *= s;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) gx[k] += __shfl_xor(gx[k], o);
    if (l == 0 && tid / 16 < M) {
        g_x[m * 3 + 0] = gx[0]; g_x[m * 3 + 1] = gx[1]; g_x[m * 3 + 2] = gx[2];
    }
}

extern "C" size_t nefes_hashgrid_table_entries(const NefesHashGridDesc* desc) {
    HgGeom g;
    uint64_t total = 0;
    if (hg_geometry(desc, &g, &total)) return 0;
    return (size_t)total;
}

extern "C" int nefes_hashgrid_fwd(const NefesHashGridDesc* desc, const float* table, int64_t M, const float* x, float* enc,
                                  void* stream) {
    if (!table || !x || !enc || M <= 0) return NEFES_E_BADARG;
    HgGeom g;
    int rc = hg_geometry(desc, &g, nullptr);
    if (rc) return rc;
    const long long n = (long long)M * g.n_levels;
    hipLaunchKernelGGL(hashgrid_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (const float2*)table, (long long)M, x, (float2*)enc);
    return (int)hipGetLastError();
}

extern "C" int nefes_hashgrid_bwd_x(const NefesHashGridDesc* desc, const float* table, int64_t M, const float* x,
                                    const float* g_enc, float* g_x, void* stream) {
    if (!table || !x || !g_enc || !g_x || M <= 0) return NEFES_E_BADARG;
    HgGeom g;
    int rc = hg_geometry(desc, &g, nullptr);
    if (rc) return rc;
    const long long n = (long long)M * 16;
    hipLaunchKernelGGL(hashgrid_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (const float2*)table, (long long)M, x, (const float2*)g_enc, g_x);
    return (int)hipGetLastError();
}
