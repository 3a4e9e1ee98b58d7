// Fused field backward-to-inputs (frozen weights: dX only, no dW), one launch.
// What autograd does in the reference for d raw -> d pts, d viewdirs through NeRFH_NFF.forward and
// Embedder.embed (script/models/nerfh_nff.py:525-576, :234-270), restated as a chain of W^T products
// on v_mfma_f32_32x32x2_f32: A operand = W^T fragments (LDS-DMA ring), B operand = the upstream
// gradient vector held in registers, ReLU derivative from the 1-bit masks the forward pass stored.
// Head activation derivatives are recovered from the raw outputs (softplus' = 1 - exp(-y), sigmoid' = y(1-y)).
#include "field_common.h"
#include "../../include/nefes_hip.h"

struct FieldBwdArgs {
    const char* stream;
    uint32_t n_slabs;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    const float* viewdirs;
    const float* raw_t;     // [N][R][S] forward output
    const float* g_raw_t;   // [N][R][S] upstream gradient
    const uint32_t* masks;  // [tiles32][MW][64]
    float* g_pts;           // [M,3]
    float* g_vs;            // [M,3] per-sample d viewdirs
    int N, S, R, C;
    long long M;
    int n_tiles;
};

template <int W, int C3>   // C3 = 3 + C
__global__ __launch_bounds__(256, 1) void field_bwd_kernel(FieldBwdArgs a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128);
    constexpr int WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;   // mask words per trunk / half-width layer
    constexpr int MW_TRUNK = 8 * WT;
    constexpr int KR = (C3 + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    WeightRing ring;
    ring.init(a.stream, a.n_slabs, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, wave, lane);
    const char* ring_lane = smem + lane * 16;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long m_raw = (long long)tile * 128 + wave * 32 + j;
        const bool valid = m_raw < a.M;
        const long long m = valid ? m_raw : a.M - 1;
        const int ray = (int)(m / a.S);
        const int smp = (int)(m - (long long)ray * a.S);
        const uint32_t* mk = a.masks + ((size_t)((valid ? m_raw : (a.M - 1)) >> 5) * MW) * 64 + lane;
        const size_t chan0 = (size_t)ray * a.R * a.S + smp;   // + ch*S
        auto RAW = [&](int ch) { return a.raw_t[chan0 + (size_t)ch * a.S]; };
        auto GRAW = [&](int ch) { return valid ? a.g_raw_t[chan0 + (size_t)ch * a.S] : 0.f; };

        float Tv[GS], Gv[GS];
        f32x16 acc2[NTH];
        uint32_t bh[WH];
        // ---- transient heads^T: 5 pre-activation gradients in compact slots (2s+h) ----
        {
            const int cT = C3 + 1;   // transient rgb channels start
            float dth[3];
            if (h == 0) {
                const float c0 = RAW(cT), c2 = RAW(cT + 2), bt = RAW(cT + 4);
                dth[0] = GRAW(cT) * (c0 * (1.f - c0));
                dth[1] = GRAW(cT + 2) * (c2 * (1.f - c2));
                dth[2] = GRAW(cT + 4) * (1.f - expf(-bt));
            } else {
                const float c1 = RAW(cT + 1), st = RAW(cT + 3);
                dth[0] = GRAW(cT + 1) * (c1 * (1.f - c1));
                dth[1] = GRAW(cT + 3) * (1.f - expf(-st));
                dth[2] = 0.f;
            }
            zero_init<NTH>(acc2);
            mma_segment<NTH, 3>(ring, ring_lane, dth, acc2);
#pragma unroll
            for (int w = 0; w < WH; ++w) bh[w] = mk[(MW_TRUNK + 3 * WH + w) * 64];   // mask of transient_encoding.4
            mask_store<NTH, 0>(Tv, acc2, bh);
        }
        // ---- transient_encoding.4^T, .2^T ----
#pragma unroll 1
        for (int tl = 2; tl >= 1; --tl) {
            zero_init<NTH>(acc2);
            mma_segment<NTH, GS>(ring, ring_lane, Tv, acc2);
#pragma unroll
            for (int w = 0; w < WH; ++w) bh[w] = mk[(MW_TRUNK + tl * WH + w) * 64];
            mask_store<NTH, 0>(Tv, acc2, bh);
        }
        // ---- static_rgb^T: 3+C gradients, compact slots ----
        {
            float dr[KR];
#pragma unroll
            for (int s = 0; s < KR; ++s) dr[s] = (2 * s + h < C3) ? GRAW(2 * s + h) : 0.f;
            zero_init<NTH>(acc2);
            mma_segment<NTH, KR>(ring, ring_lane, dr, acc2);
#pragma unroll
            for (int w = 0; w < WH; ++w) bh[w] = mk[(MW_TRUNK + w) * 64];   // mask of dir_encoding
            mask_store<NTH, 0>(Gv, acc2, bh);
        }
        float H[HS];
        float dDv[16];
        {
            // ---- [transient_encoding.0 ; dir_encoding]^T -> d final (NTW tiles) + d dir-embedding (1 tile) ----
            f32x16 acc9[NTW + 1];
            zero_init<NTW + 1>(acc9);
            mma_segment<NTW + 1, GS>(ring, ring_lane, Tv, acc9);
            mma_segment<NTW + 1, GS>(ring, ring_lane, Gv, acc9);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) H[t * 16 + r] = acc9[t][r];
#pragma unroll
            for (int r = 0; r < 16; ++r) dDv[r] = acc9[NTW][r];
        }
        f32x16 acc[NTW];
        uint32_t bt[WT];
        {
            // ---- xyz_encoding_final^T + static_sigma^T (one extra k-step) -> d h8 ----
            float dsg[1];
            const float sg = RAW(C3);
            dsg[0] = h == 0 ? GRAW(C3) * (1.f - expf(-sg)) : 0.f;
            zero_init<NTW>(acc);
            mma_segment<NTW, HS>(ring, ring_lane, H, acc);
            mma_segment<NTW, 1>(ring, ring_lane, dsg, acc);
#pragma unroll
            for (int w = 0; w < WT; ++w) bt[w] = mk[(7 * WT + w) * 64];
            mask_store<NTW, 0>(H, acc, bt);
        }
        f32x16 accE[2];
        zero_init<2>(accE);
        // ---- xyz_encoding_8^T .. xyz_encoding_2^T; layer 5 also emits the skip's d embedding ----
#pragma unroll 1
        for (int l = 8; l >= 2; --l) {
            if (l == 5) {
                f32x16 acc10[NTW + 2];
                zero_init<NTW + 2>(acc10);
                mma_segment<NTW + 2, HS>(ring, ring_lane, H, acc10);
                accE[0] = acc10[0];
                accE[1] = acc10[1];
#pragma unroll
                for (int w = 0; w < WT; ++w) bt[w] = mk[(3 * WT + w) * 64];
                mask_store<NTW, 2>(H, acc10, bt);
            } else {
                zero_init<NTW>(acc);
                mma_segment<NTW, HS>(ring, ring_lane, H, acc);
#pragma unroll
                for (int w = 0; w < WT; ++w) bt[w] = mk[((l - 2) * WT + w) * 64];
                mask_store<NTW, 0>(H, acc, bt);
            }
        }
        // ---- xyz_encoding_1^T accumulates onto the skip's d embedding ----
        mma_segment<2, HS>(ring, ring_lane, H, accE);

        // ---- embedding backward (Embedder.embed :257-267) ----
        float x[3], v[3];
        if (a.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = a.pts[m * 3 + c];
        } else {
            const float zz = a.z[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = add_rn(a.rays_o[ray * 3 + c], mul_rn(a.rays_d[ray * 3 + c], zz));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[ray * 3 + c];
        float dE[32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dE[t * 16 + r] = accE[t][r];
        float gx[3], gv[3];
        embed_slots_bwd<NEFES_N_FREQ_XYZ>(gx, dE, x, h);
        embed_slots_bwd<NEFES_N_FREQ_DIR>(gv, dDv, v, h);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gx[c] += __shfl_xor(gx[c], 32);
            gv[c] += __shfl_xor(gv[c], 32);
        }
        if (valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.g_pts[m * 3 + c] = gx[c];
                a.g_vs[m * 3 + c] = gv[c];
            }
        }
    }
    ring.drain();
}

template <int W, int C3>
static int launch_bwd(const FieldBwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_RING_SLOTS * NEFES_SLAB_BYTES;
    auto k = field_bwd_kernel<W, C3>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = a.n_tiles < cus ? a.n_tiles : cus;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

extern "C" int nefes_field_bwd(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                               const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                               const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                               float* g_viewdirs_s, void* stream) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !g_pts || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    if (!pts && !(rays_o && rays_d && z)) return NEFES_E_BADARG;
    if (!desc->has_transient) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[NEFES_STREAM_BWD_FULL];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldBwdArgs a;
    a.stream = (const char*)packed + si.slab_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + 6;
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    if (desc->width == 256 && desc->feat_dim == 16) return launch_bwd<256, 19>(a, st);
    if (desc->width == 128 && desc->feat_dim == 128) return launch_bwd<128, 131>(a, st);
    return NEFES_E_UNSUPPORTED;
}
