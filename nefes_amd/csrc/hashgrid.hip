// Multiresolution hash-grid positional encoding (BASELINE config 4) and its backward to x.
// The reference only configures this encoding (script/models/nerfh_tcnn.py:60-75, input normalisation :151-156);
// the arithmetic is tiny-cuda-nn's published algorithm, restated in oracle/hashgrid_ref.py (PARITY UNPINNED:
// tiny-cuda-nn is neither vendored nor version-pinned by the reference, and the model is orphaned there).
//
// Roofline: gather-bound.  Per sample: 16 levels x 8 corners x 8 B (fp32 x2) = 1 KiB of table reads, 12 B in, 128 B out.
// The 48.8 MB table stays in the 256 MiB Infinity Cache.  One thread per (sample, level): a wave covers 4 samples,
// writes 512 contiguous bytes, and the backward reduces the 16 levels of a sample with 4 xor-shuffles.
#include "hashgrid.h"

__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(HgGeom g, const float2* __restrict__ table, long long M,
                                                           const float* __restrict__ x, float2* __restrict__ enc) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = tid / g.n_levels;
    const int l = (int)(tid - m * g.n_levels);
    if (m >= M) return;
    const HgLevel L = g.lv[l];
    const float xs[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
    const float2 acc = hg_level_fwd(L, g.bound, table, xs);
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
        const float xs[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
        const float* gep = (const float*)&g_enc[m * g.n_levels + l];                       // (read once: non-temporal)
        const float2 ge = make_float2(__builtin_nontemporal_load(gep), __builtin_nontemporal_load(gep + 1));
        hg_level_bwd_x(L, g.bound, g.inv_range, table, xs, ge, gx);
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
