// Fused field backward-to-inputs with the hidden products as fp16 two-part split products (field_h3.h): what autograd does in
// the reference for d raw -> d pts, d viewdirs through NeRFH_NFF.forward and Embedder.embed (script/models/nerfh_nff.py:525-576,
// :234-270) with frozen weights.  Same chain of W^T products, same inputs / outputs / ReLU-mask words as field_bwd_kernel
// (field_bwd.hip); the transposed products of transient_encoding.{4,2,0}, dir_encoding, xyz_encoding_final and layers 8..1 run on
// v_mfma_f32_32x32x16_f16 as hh + hl + lh of power-of-two scaled (hi, lo) fp16 pairs; the three narrow head products (3+C, 5 and
// 1 k-values) stay on the fp32 MFMA.  Every gradient vector carries a per-lane scale exponent (field_h3.h).
#define NEFES_SLAB_KIB NEFES_H3_BWD_SLAB_KIB
#define NEFES_B_BATCH 2
#define NEFES_B_BATCH_NT8 4
#include "field_common.h"
#include "field_x6.h"
#include "field_h3.h"
#include "../../include/nefes_hip.h"
#define NEFES_H3B_SLOTS 3   // 96 KiB ring (+ the tile's ReLU masks and the exponent table)

struct FieldBwdH3Args {
    const char* stream;
    const int* wexp;        // weight-scale exponent per segment (layout.h NEFES_H3B_*), in the blob
    uint32_t n_slabs;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    const float* viewdirs;
    const float* raw_t;     // [N][R][S] forward output
    const float* g_raw_t;   // [N][R][S] upstream gradient
    const uint32_t* masks;  // [tiles32][MW][64]
    float* g_pts;           // [M,3] (NEFES_XYZ_FREQ10)
    float* g_enc;           // [M,32] (NEFES_XYZ_EXTERNAL32)
    float* g_vs;            // [M,3] per-sample d viewdirs
    int N, S, R, C;
    long long M;
    int n_tiles;
};

// largest magnitude of accumulator tiles [T0, T0 + NT) (outputs of the fp32 head products; the fp16 products report theirs)
template <int NT, int T0, int NX>
__device__ __forceinline__ float tiles_absmax(const f32x16 (&X)[NX]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) tile_max_acc<2>(m, X[T0 + t]);
    return m;
}

template <int W, int C3, int ENC>   // C3 = 3 + C; ENC = NEFES_XYZ_*
__global__ __launch_bounds__(256, 1) void field_bwd_h3_kernel(FieldBwdH3Args a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128);
    constexpr int WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;   // mask words per trunk / half-width layer
    constexpr int MW_TRUNK = 8 * WT;
    constexpr int KR = (C3 + 1) / 2;
    static_assert(MW % 4 == 0, "mask words are staged as 16-byte groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    int* wexp = (int*)(smem + NEFES_H3B_SLOTS * NEFES_SLAB_BYTES + (size_t)4 * (MW + 8) * 256);
    if (threadIdx.x < NEFES_H3B_N) wexp[threadIdx.x] = a.wexp[threadIdx.x];
    WeightRing<NEFES_H3B_SLOTS> ring;
    ring.init(a.stream, a.n_slabs, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, wave, lane);
    __syncthreads();
    const char* ring_lane = smem + lane * 16;
    ring.prime(ring_lane);
    // this wave's mask words in LDS: [MW/4][64 lanes][4 words]
    uint32_t* mlds = (uint32_t*)(smem + NEFES_H3B_SLOTS * NEFES_SLAB_BYTES) + wave * ((MW + 8) * 64) + lane * 4;
    auto MASKW = [&](int w) { return mlds[(w >> 2) * 256 + (w & 3)]; };
    float* stash = (float*)mlds;                 // words [MW, MW+8) of the same per-lane LDS column: x, v, d sigma
    auto STASH = [&](int k) -> float& { return stash[((MW + k) >> 2) * 256 + ((MW + k) & 3)]; };

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long m_raw = (long long)tile * 128 + wave * 32 + j;
        const bool valid = m_raw < a.M;
        const long long m = valid ? m_raw : a.M - 1;
        const int ray = (int)(m / a.S);
        const int smp = (int)(m - (long long)ray * a.S);
        const size_t chan0 = (size_t)ray * a.R * a.S + smp;   // + ch*S

        // ================= all global loads of the tile, then ONE explicit completion point =================
        float in_o[3] = {0.f, 0.f, 0.f}, in_d[3] = {0.f, 0.f, 0.f}, in_z = 0.f, v[3];
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            // the gradient w.r.t. the supplied embedding does not depend on the sample position
        } else if (a.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) in_o[c] = a.pts[m * 3 + c];
        } else {
            in_z = a.z[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) { in_o[c] = a.rays_o[ray * 3 + c]; in_d[c] = a.rays_d[ray * 3 + c]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[ray * 3 + c];
        const int cT = C3 + 1;                       // transient rgb channels start
        float y_th[3] = {0.f, 0.f, 0.f}, g_th[3] = {0.f, 0.f, 0.f}, y_sg, g_sg, dr[KR];
#pragma unroll
        for (int s = 0; s < 3; ++s) {            // compact slot (s,h) <-> transient-head row 2s+h (5 rows)
            const int row = 2 * s + h;
            const int ch = cT + (row < 5 ? row : 4);
            y_th[s] = a.raw_t[chan0 + (size_t)ch * a.S];
            g_th[s] = a.g_raw_t[chan0 + (size_t)ch * a.S];
        }
        y_sg = a.raw_t[chan0 + (size_t)C3 * a.S];
        g_sg = a.g_raw_t[chan0 + (size_t)C3 * a.S];
#pragma unroll
        for (int s = 0; s < KR; ++s) {               // compact slot (s,h) <-> static rgb/feature channel 2s+h
            const int ch = 2 * s + h;
            dr[s] = a.g_raw_t[chan0 + (size_t)(ch < C3 ? ch : C3 - 1) * a.S];
        }
        uint4 mq[MW / 4];
        {
            const uint32_t* mk32 = a.masks + ((size_t)(m >> 5) * MW) * 64 + lane;
#pragma unroll
            for (int q = 0; q < MW / 4; ++q) {
                mq[q].x = mk32[(4 * q + 0) * 64]; mq[q].y = mk32[(4 * q + 1) * 64];
                mq[q].z = mk32[(4 * q + 2) * 64]; mq[q].w = mk32[(4 * q + 3) * 64];
            }
        }
        loads_landed();
        pin(in_o); pin(in_d); pin(in_z); pin(v); pin(y_th); pin(g_th); pin(y_sg); pin(g_sg); pin(dr);
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) { pin(mq[q].x); pin(mq[q].y); pin(mq[q].z); pin(mq[q].w); }
        // ======================================================================================================
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) *(uint4*)(mlds + q * 256) = mq[q];     // own lane's words only: no barrier needed
        if (!valid) {
#pragma unroll
            for (int s = 0; s < 3; ++s) g_th[s] = 0.f;
            g_sg = 0.f;
#pragma unroll
            for (int s = 0; s < KR; ++s) dr[s] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < KR; ++s) dr[s] = (2 * s + h < C3) ? dr[s] : 0.f;
        // head activation derivatives from the outputs: sigmoid' = y(1-y), softplus' = 1 - exp(-y)
        float dth[3];
        if (h == 0) {   // rows 0 (rgb_t0), 2 (rgb_t2), 4 (beta)
            dth[0] = g_th[0] * (y_th[0] * (1.f - y_th[0]));
            dth[1] = g_th[1] * (y_th[1] * (1.f - y_th[1]));
            dth[2] = g_th[2] * (1.f - expf(-y_th[2]));
        } else {        // rows 1 (rgb_t1), 3 (sigma_t), pad
            dth[0] = g_th[0] * (y_th[0] * (1.f - y_th[0]));
            dth[1] = g_th[1] * (1.f - expf(-y_th[1]));
            dth[2] = 0.f;
        }
        // values needed only at the end of the tile wait in LDS, not in registers
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            STASH(c) = a.pts ? in_o[c] : add_rn(in_o[c], mul_rn(in_d[c], in_z));
            STASH(3 + c) = v[c];
        }
        STASH(6) = h == 0 ? g_sg * (1.f - expf(-y_sg)) : 0.f;

        auto load_bits = [&](uint32_t* b, int word0, int n) {
            for (int w = 0; w < n; ++w) b[w] = MASKW(word0 + w);
        };
        uint32_t bh[WH], bt[WT];
        float mx[2], mx_[2];
        // exponent for the next operand from the lane pair's largest magnitude, capped so that the output scale stays finite
        auto next_exp = [&](float m_lane, int es_in, int ew) { return cap_exp(pick_exp(pair_max(m_lane)), es_in, ew); };

        f32x16 G2[NTH], T3[NTH], T4[NTH];
        // ---- static_rgb^T (fp32): 3+C gradients in compact slots -> d(dir_encoding output), scale 2^0 ----
        mma_run<NTH, KR, 0, true>(ring, ring_lane, ArrayIn<KR>{dr}, ZeroInit{}, G2);
        // ---- transient heads^T (fp32): 5 pre-activation gradients -> d(transient_encoding.4 output), scale 2^0 ----
        mma_run<NTH, 3, 0, true>(ring, ring_lane, ArrayIn<3>{dth}, ZeroInit{}, T3);
        // ---- transient_encoding.4^T, .2^T ----
        int es4, es3;
        {
            load_bits(bh, MW_TRUNK + 3 * WH, WH);
            const int ew = wexp[NEFES_H3B_T2], ex = next_exp(tiles_absmax<NTH, 0>(T3), 0, ew);
            es4 = ex + ew;
            mma_run_h3<NTH, GS / 8, 0, true, 2>(ring, ring_lane, MaskedSplitH<NTH, WH, 0>{T3, bh, pow2i(ex)}, ZeroInit{}, T4, mx);
        }
        {
            load_bits(bh, MW_TRUNK + 2 * WH, WH);
            const int ew = wexp[NEFES_H3B_T1], ex = next_exp(mx[0], es4, ew);
            es3 = es4 + ex + ew;
            mma_run_h3<NTH, GS / 8, 0, true, 2>(ring, ring_lane, MaskedSplitH<NTH, WH, 0>{T4, bh, pow2i(ex)}, ZeroInit{}, T3, mx);
        }
        // Full-width accumulators, ping-pong.  Tiles [2, NTW+2) hold a layer's d hidden; XA tile 1 = d dir-embedding;
        // XB tiles 0,1 = d xyz-embedding (written by layer 5, accumulated by layer 1).
        f32x16 XA[NTW + 2], XB[NTW + 2];
        // ---- [transient_encoding.0 ; dir_encoding]^T -> d dir-embedding (tile 1) + d final (tiles 2..): both products
        //      accumulate into the same tiles, so both operands are brought to one common scale 2^tau ----
        int es_dt;
        {
            const int ew = wexp[NEFES_H3B_T0];                                    // = wexp[NEFES_H3B_DIR]: one scale for the pair (pack.cpp)
            const int tau_t = es3 + pick_exp(pair_max(mx[0]));                    // what d(transient_encoding.0 output) could carry
            const int tau_g = pick_exp(pair_max(tiles_absmax<NTH, 0>(G2)));       // what d(dir_encoding output) could carry (scale 2^0)
            int tau = tau_t < tau_g ? tau_t : tau_g;
            tau = tau < 100 - ew ? tau : 100 - ew;
            es_dt = tau + ew;
            load_bits(bh, MW_TRUNK + WH, WH);
            mma_run_h3<NTW + 1, GS / 8, 1, true>(ring, ring_lane, MaskedSplitH<NTH, WH, 0>{T3, bh, pow2i(tau - es3)}, ZeroInit{}, XA, mx_);
            load_bits(bh, MW_TRUNK, WH);
            mma_run_h3<NTW + 1, GS / 8, 1, false, 2, 1>(ring, ring_lane, MaskedSplitH<NTH, WH, 0>{G2, bh, pow2i(tau)}, ZeroInit{}, XA, mx);
        }
        // ---- xyz_encoding_final^T (no ReLU on its output) + static_sigma^T (one extra fp32 k-step) -> d h8 ----
        int es_b;
        {
            const int ew = wexp[NEFES_H3B_FINAL], ex = next_exp(mx[1], es_dt, ew);   // mx[1]: the d final tiles (tile 1 is the dir part)
            es_b = es_dt + ex + ew;
            mma_run_h3<NTW, W / 16, 2, true>(ring, ring_lane, IdentSplitH<NTW + 2, 2>{XA, pow2i(ex)}, ZeroInit{}, XB, mx_);
            float dsg[1];
            dsg[0] = STASH(6) * pow2i(es_b);
            mma_run<NTW, 1, 2, false>(ring, ring_lane, ArrayIn<1>{dsg}, ZeroInit{}, XB);
            mx[0] = tiles_absmax<NTW, 2>(XB);
        }
        // ---- xyz_encoding_8^T .. xyz_encoding_2^T, straight-line (XB -> XA -> XB ...).  Layer 5 also emits the skip's d
        //      xyz-embedding into XB tiles 0,1 (reported apart: mx[0]; the d hidden tiles: mx[1]). ----
        int es_a = 0, es_e = 0;
#define NEFES_BWD_LAYER(L, SRC, DST, ES_SRC, ES_DST, NTILES, T0, MSRC)                                                 \
        {                                                                                                           \
            load_bits(bt, ((L) - 1) * WT, WT);                                                                      \
            const int ew = wexp[NEFES_H3B_L8 + 8 - (L)], ex = next_exp(MSRC, ES_SRC, ew);                           \
            ES_DST = ES_SRC + ex + ew;                                                                              \
            mma_run_h3<NTILES, W / 16, T0, true, 2, 2 - (T0)>(ring, ring_lane, MaskedSplitH<NTW + 2, WT, 2>{SRC, bt, pow2i(ex)}, ZeroInit{}, DST, mx); \
        }
        NEFES_BWD_LAYER(8, XB, XA, es_b, es_a, NTW, 2, mx[0])
        NEFES_BWD_LAYER(7, XA, XB, es_a, es_b, NTW, 2, mx[1])
        NEFES_BWD_LAYER(6, XB, XA, es_b, es_a, NTW, 2, mx[1])
        NEFES_BWD_LAYER(5, XA, XB, es_a, es_b, NTW + 2, 0, mx[1])
        es_e = es_b;                                                         // scale of the d xyz-embedding tiles XB[0], XB[1]
        NEFES_BWD_LAYER(4, XB, XA, es_b, es_a, NTW, 2, mx[1])
        NEFES_BWD_LAYER(3, XA, XB, es_a, es_b, NTW, 2, mx[1])
        NEFES_BWD_LAYER(2, XB, XA, es_b, es_a, NTW, 2, mx[1])
#undef NEFES_BWD_LAYER
        // ---- xyz_encoding_1^T accumulates onto the skip's d embedding: bring those two tiles to the new product's scale ----
        int es_1;
        {
            load_bits(bt, 0, WT);
            const int ew = wexp[NEFES_H3B_L1], ex = next_exp(mx[1], es_a, ew);
            es_1 = es_a + ex + ew;
            int de = es_1 - es_e;
            de = de < -120 ? -120 : (de > 120 ? 120 : de);
            const float resc = pow2i(de);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) XB[t][r] *= resc;
            mma_run_h3<2, W / 16, 0, false>(ring, ring_lane, MaskedSplitH<NTW + 2, WT, 2>{XA, bt, pow2i(ex)}, ZeroInit{}, XB, mx_);
        }
        float dDv[16];
        {
            const float inv = pow2i(-es_dt);
#pragma unroll
            for (int r = 0; r < 16; ++r) dDv[r] = XA[1][r] * inv;
        }

        // ---- embedding backward (Embedder.embed :257-267) ----
        float x[3], gx[3] = {0.f, 0.f, 0.f}, gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = STASH(c); v[c] = STASH(3 + c); }
        const float inv1 = pow2i(-es_1);
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            float ge[NEFES_X_STEPS];
#pragma unroll
            for (int s = 0; s < NEFES_X_STEPS; ++s) ge[s] = XB[0][s] * inv1;
            if (valid) {
                float* gp = a.g_enc + m * 32 + h;
#pragma unroll
                for (int s = 0; s < NEFES_X_STEPS; ++s) gp[2 * s] = ge[s];
            }
        } else {
            float dE[32];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dE[t * 16 + r] = XB[t][r] * inv1;
            embed_slots_bwd<NEFES_N_FREQ_XYZ>(gx, dE, x, h);
        }
        embed_slots_bwd<NEFES_N_FREQ_DIR>(gv, dDv, v, h);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gx[c] += __shfl_xor(gx[c], 32);
            gv[c] += __shfl_xor(gv[c], 32);
        }
        if (valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (ENC != NEFES_XYZ_EXTERNAL32) a.g_pts[m * 3 + c] = gx[c];
                a.g_vs[m * 3 + c] = gv[c];
            }
        }
    }
    ring.drain();
}

template <int W, int C3, int ENC>
static int launch_bwd_h3(const FieldBwdH3Args& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_H3B_SLOTS * NEFES_SLAB_BYTES + (size_t)4 * (8 * (W / 64) + 4 * (W / 128) + 8) * 256 + 256;
    auto k = field_bwd_h3_kernel<W, C3, ENC>;
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

// Instances spread over two objects built from this one source (Makefile: -DNEFES_TU_PART=0..1): part 0 = entry point + the
// Wd = 256 frequency-embedding instance, part 1 = hash-grid and Wd = 128 instances.
#ifndef NEFES_TU_PART
#define NEFES_TU_PART 0
#endif
enum { BWD_H3_256_EXT = 0, BWD_H3_128 };
int nefes_bwd_h3_launch_part1(int which, const FieldBwdH3Args& a, hipStream_t st);

#if NEFES_TU_PART == 1
int nefes_bwd_h3_launch_part1(int which, const FieldBwdH3Args& a, hipStream_t st) {
    switch (which) {
        case BWD_H3_256_EXT: return launch_bwd_h3<256, 19, NEFES_XYZ_EXTERNAL32>(a, st);
        case BWD_H3_128: return launch_bwd_h3<128, 131, NEFES_XYZ_FREQ10>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#else   // part 0

extern "C" int nefes_field_bwd_h3(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                  const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                  float* g_xyz_enc, float* g_viewdirs_s, void* stream) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    if (ext ? !g_xyz_enc : (!g_pts || (!pts && !(rays_o && rays_d && z)))) return NEFES_E_BADARG;
    if (!desc->has_transient) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[NEFES_STREAM_BWD_FULL_H3];
    if (si.n_slabs == 0 || si.scale_count < NEFES_H3B_N) return NEFES_E_UNSUPPORTED;
    FieldBwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.wexp = (const int*)((const char*)packed + si.bias_off) + si.scale_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_enc = g_xyz_enc; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + 6;
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    if (desc->width == 256 && desc->feat_dim == 16 && !ext) return launch_bwd_h3<256, 19, NEFES_XYZ_FREQ10>(a, st);
    if (desc->width == 256 && desc->feat_dim == 16 && ext) return nefes_bwd_h3_launch_part1(BWD_H3_256_EXT, a, st);
    if (desc->width == 128 && desc->feat_dim == 128 && !ext) return nefes_bwd_h3_launch_part1(BWD_H3_128, a, st);
    return NEFES_E_UNSUPPORTED;
}
#endif   // NEFES_TU_PART
