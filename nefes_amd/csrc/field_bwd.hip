// Fused field backward-to-inputs (frozen weights: dX only, no dW), one launch.
// What autograd does in the reference for d raw -> d pts, d viewdirs through NeRFH_NFF.forward and
// Embedder.embed (script/models/nerfh_nff.py:525-576, :234-270), restated as a chain of W^T products
// on v_mfma_f32_32x32x2_f32: A operand = W^T fragments (LDS-DMA ring), B operand = the upstream
// gradient vector held in registers, ReLU derivative from the 1-bit masks the forward pass stored.
// Head activation derivatives are recovered from the raw outputs (softplus' = 1 - exp(-y), sigmoid' = y(1-y)).
#define NEFES_SLAB_KIB NEFES_BWD_SLAB_KIB
// look-ahead batches of 2 k-steps for the narrow segments: with 4 the Wd = 128 bf16x6 instance spills (316 B scratch)
#define NEFES_B_BATCH 2
#define NEFES_B_BATCH_NT8 4
// every run of compiler-placed MFMAs ends with field_common.h mfma_results_fence: hipcc pads an MFMA's result against its own vector
// instructions, but takes the first path it finds to the MFMA where two join (seen: 7 of 18 wait states on the path that skips the
// mask stores) and does not look into the asm statements of the operand functors at all (tools/hazard_lint.py rules B1 / B2)
#define NEFES_ASM_READS_ACC
#include "field_common.h"
#include "field_x6.h"
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
    float* g_pts;           // [M,3] (NEFES_XYZ_FREQ10)
    float* g_enc;           // [M,32] (NEFES_XYZ_EXTERNAL32)
    float* g_vs;            // [M,3] per-sample d viewdirs
    int N, S, R, C;
    long long M;
    int n_tiles;
    float* dacts;           // TRAIN instances: [n_tiles][rows][128] gradient buffer (layout.h row map)
    int rows;
};


// X6: the eight 256x256 transposed products (xyz_encoding_final^T, layers 8..2) as bf16x6 split products (field_x6.h);
// the stream then is NEFES_STREAM_BWD_FULL_X6.  Same mask words, same outputs.
// HAS_T = false: the static head only (NEFES_FIELD_STATIC forward: raw channels rgb+feature, sigma); the stream then is
// NEFES_STREAM_BWD_STATIC and the transient segments are skipped.
// TRAIN: every gradient vector a product consumes (= d loss / d pre-activation of a hidden layer, ReLU mask applied) is
// also written to a.dacts, where the weight-gradient kernel (train.hip) reads it: one fused launch replaces the
// layer-by-layer nefes_train_dx chain.
template <int W, int C3, int ENC, int X6 = 0, bool HAS_T = true, bool TRAIN = false>   // C3 = 3 + C; ENC = NEFES_XYZ_*; X6 = 0, 6 or 3 products
__global__ __launch_bounds__(256, 1) void field_bwd_kernel(FieldBwdArgs a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128);
    constexpr int WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;   // mask words per trunk / half-width layer
    constexpr int MW_TRUNK = 8 * WT;
    constexpr int KR = (C3 + 1) / 2;
    static_assert(MW % 4 == 0, "mask words are staged as 16-byte groups");
    constexpr int NP = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    WeightRing<NEFES_BWD_SLOTS> ring;
    ring.init(a.stream, a.n_slabs, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, wave, lane);
    const char* ring_lane = smem + lane * 16;
    ring.prime(ring_lane);
    // this wave's mask words in LDS: [MW/4][64 lanes][4 words]
    uint32_t* mlds = (uint32_t*)(smem + NEFES_BWD_SLOTS * NEFES_SLAB_BYTES) + wave * ((MW + 8) * 64) + lane * 4;
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
        // forward outputs needed for the head activation derivatives, and the upstream gradient (this lane half's slots)
        const int cT = C3 + 1;                       // transient rgb channels start
        float y_th[3] = {0.f, 0.f, 0.f}, g_th[3] = {0.f, 0.f, 0.f}, y_sg, g_sg, dr[KR];
        if constexpr (HAS_T) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {            // compact slot (s,h) <-> transient-head row 2s+h (5 rows)
                const int row = 2 * s + h;
                const int ch = cT + (row < 5 ? row : 4);
                y_th[s] = a.raw_t[chan0 + (size_t)ch * a.S];
                g_th[s] = a.g_raw_t[chan0 + (size_t)ch * a.S];
            }
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
            const uint4* mk = (const uint4*)nullptr;
            const uint32_t* mk32 = a.masks + ((size_t)(m >> 5) * MW) * 64 + lane;
#pragma unroll
            for (int q = 0; q < MW / 4; ++q) {
                mq[q].x = mk32[(4 * q + 0) * 64]; mq[q].y = mk32[(4 * q + 1) * 64];
                mq[q].z = mk32[(4 * q + 2) * 64]; mq[q].w = mk32[(4 * q + 3) * 64];
            }
            (void)mk;
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

        // Consumer-side masking (see field_common.h): every product reads its upstream gradient straight out of the
        // producer's accumulators and applies that layer's ReLU mask on the way in; nothing is copied between layers.
        auto load_bits = [&](uint32_t* b, int word0, int n) {
            for (int w = 0; w < n; ++w) b[w] = MASKW(word0 + w);
        };
        uint32_t bh[WH], bt[WT];
        // TRAIN: this lane's column of a hidden block in the gradient buffer (first row + 4 rows for lane half 1)
        auto gptr = [&](int block) -> float* {
            if constexpr (!TRAIN) return nullptr;
            else return a.dacts + (size_t)tile * a.rows * 128 + (size_t)(nefes_train_row(W, 0, block) >> 5) * 4096 + nefes_train_lane_off(wave, j, h);
        };
        f32x16 G2[NTH], T3[NTH], T4[NTH];
        // ---- static_rgb^T: 3+C gradients in compact slots -> d(dir_encoding output) ----
        mma_run<NTH, KR, 0, true>(ring, ring_lane, ArrayIn<KR>{dr}, ZeroInit{}, G2);
        if constexpr (HAS_T) {
        // ---- transient heads^T: 5 pre-activation gradients -> d(transient_encoding.4 output) ----
        mma_run<NTH, 3, 0, true>(ring, ring_lane, ArrayIn<3>{dth}, ZeroInit{}, T3);
        // ---- transient_encoding.4^T, .2^T ----
        load_bits(bh, MW_TRUNK + 3 * WH, WH);
        if constexpr (X6) mma_run_x6<NTH, GS / 8, 0, true, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTH, WH, 0>{T3, bh}, gptr(NEFES_TB_T2)), ZeroInit{}, T4);
        else mma_run<NTH, GS, 0, true>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTH, WH>{T3, bh}, gptr(NEFES_TB_T2)), ZeroInit{}, T4);
        load_bits(bh, MW_TRUNK + 2 * WH, WH);
        if constexpr (X6) mma_run_x6<NTH, GS / 8, 0, true, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTH, WH, 0>{T4, bh}, gptr(NEFES_TB_T1)), ZeroInit{}, T3);
        else mma_run<NTH, GS, 0, true>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTH, WH>{T4, bh}, gptr(NEFES_TB_T1)), ZeroInit{}, T3);
        }
        // Full-width accumulators, ping-pong.  Tiles [2, NTW+2) hold a layer's d hidden; XA tile 1 = d dir-embedding;
        // XB tiles 0,1 = d xyz-embedding (written by layer 5, accumulated by layer 1).
        f32x16 XA[NTW + 2], XB[NTW + 2];
        // ---- [transient_encoding.0 ; dir_encoding]^T -> d dir-embedding (tile 1) + d final (tiles 2..) ----
        if constexpr (HAS_T) {
            load_bits(bh, MW_TRUNK + WH, WH);
            if constexpr (X6) mma_run_x6<NTW + 1, GS / 8, 1, true, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTH, WH, 0>{T3, bh}, gptr(NEFES_TB_T0)), ZeroInit{}, XA);
            else mma_run<NTW + 1, GS, 1, true>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTH, WH>{T3, bh}, gptr(NEFES_TB_T0)), ZeroInit{}, XA);
        }
        load_bits(bh, MW_TRUNK, WH);
        if constexpr (X6) mma_run_x6<NTW + 1, GS / 8, 1, !HAS_T, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTH, WH, 0>{G2, bh}, gptr(NEFES_TB_DIR)), ZeroInit{}, XA);
        else mma_run<NTW + 1, GS, 1, !HAS_T>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTH, WH>{G2, bh}, gptr(NEFES_TB_DIR)), ZeroInit{}, XA);
        // ---- xyz_encoding_final^T (no ReLU on its output) + static_sigma^T (one extra k-step) -> d h8 ----
        {
            float dsg[1];
            dsg[0] = STASH(6);
            if constexpr (X6) mma_run_x6<NTW, W / 16, 2, true, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(IdentSplit<NTW + 2, 2>{XA}, gptr(NEFES_TB_FINAL)), ZeroInit{}, XB);
            else mma_run<NTW, HS, 2, true>(ring, ring_lane, wrap_store<TRAIN>(IdentIn<NTW + 2, 2>{XA}, gptr(NEFES_TB_FINAL)), ZeroInit{}, XB);
            mma_run<NTW, 1, 2, false>(ring, ring_lane, ArrayIn<1>{dsg}, ZeroInit{}, XB);
        }
        // ---- xyz_encoding_8^T .. xyz_encoding_2^T, straight-line (XB -> XA -> XB ...): a runtime loop over the ping-pong
        //      pair makes the register allocator shuffle and spill whole accumulator tiles at the back-edge.
        //      Layer 5 also emits the skip's d xyz-embedding into XB tiles 0,1. ----
#define NEFES_BWD_LAYER(L, SRC, DST, NTILES, T0)                                                            \
        load_bits(bt, ((L) - 1) * WT, WT);                                                              \
        if constexpr (X6) mma_run_x6<NTILES, W / 16, T0, true, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTW + 2, WT, 2>{SRC, bt}, gptr(NEFES_TB_L1 + (L) - 1)), ZeroInit{}, DST); \
        else mma_run<NTILES, HS, T0, true>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTW + 2, WT, 2>{SRC, bt}, gptr(NEFES_TB_L1 + (L) - 1)), ZeroInit{}, DST);
        NEFES_BWD_LAYER(8, XB, XA, NTW, 2)
        NEFES_BWD_LAYER(7, XA, XB, NTW, 2)
        NEFES_BWD_LAYER(6, XB, XA, NTW, 2)
        NEFES_BWD_LAYER(5, XA, XB, NTW + 2, 0)
        NEFES_BWD_LAYER(4, XB, XA, NTW, 2)
        NEFES_BWD_LAYER(3, XA, XB, NTW, 2)
        NEFES_BWD_LAYER(2, XB, XA, NTW, 2)
#undef NEFES_BWD_LAYER
        // ---- xyz_encoding_1^T accumulates onto the skip's d embedding ----
        load_bits(bt, 0, WT);
        if constexpr (X6) mma_run_x6<2, W / 16, 0, false, NP>(ring, ring_lane, wrap_store_x6<TRAIN>(MaskedSplit<NTW + 2, WT, 2>{XA, bt}, gptr(NEFES_TB_L1)), ZeroInit{}, XB);
        else mma_run<2, HS, 0, false>(ring, ring_lane, wrap_store<TRAIN>(MaskedIn<NTW + 2, WT, 2>{XA, bt}, gptr(NEFES_TB_L1)), ZeroInit{}, XB);
        float dDv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dDv[r] = XA[1][r];

        // ---- embedding backward (Embedder.embed :257-267) ----
        float x[3], gx[3] = {0.f, 0.f, 0.f}, gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = STASH(c); v[c] = STASH(3 + c); }
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            // tile 0, compact slots (feature 2s+h; tile 1 is padding); copied out with constant indices before the stores
            float ge[NEFES_X_STEPS];
#pragma unroll
            for (int s = 0; s < NEFES_X_STEPS; ++s) ge[s] = XB[0][s];
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
                for (int r = 0; r < 16; ++r) dE[t * 16 + r] = XB[t][r];
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

template <int W, int C3, int ENC, int X6 = 0, bool HAS_T = true, bool TRAIN = false>
static int launch_bwd(const FieldBwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_BWD_SLOTS * NEFES_SLAB_BYTES + (size_t)4 * (8 * (W / 64) + 4 * (W / 128) + 8) * 256;
    auto k = field_bwd_kernel<W, C3, ENC, X6, HAS_T, TRAIN>;
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

// The kernel instances are spread over four objects built from this one source (Makefile: -DNEFES_TU_PART=0..3), because each
// fully unrolled instance takes hipcc the better part of a minute: part 0 = entry points + the fp32 FULL instances, part 1 =
// bf16x6 instances, part 2 = three-product and static-head instances, part 3 = TRAIN instances.
#ifndef NEFES_TU_PART
#define NEFES_TU_PART 0
#endif
enum { BWD_X6_256 = 0, BWD_X6_256_EXT, BWD_X6_128, BWD_STATIC_256, BWD_STATIC_128, BWD_TRAIN_256_FULL,
       BWD_TRAIN_256_STATIC, BWD_TRAIN_128_FULL, BWD_TRAIN_128_STATIC };
int nefes_bwd_launch_part1(int which, const FieldBwdArgs& a, hipStream_t st);
int nefes_bwd_launch_part2(int which, const FieldBwdArgs& a, hipStream_t st);
int nefes_bwd_launch_part3(int which, const FieldBwdArgs& a, hipStream_t st);

#if NEFES_TU_PART == 1
int nefes_bwd_launch_part1(int which, const FieldBwdArgs& a, hipStream_t st) {
    switch (which) {
        case BWD_X6_256: return launch_bwd<256, 19, NEFES_XYZ_FREQ10, 6>(a, st);
        case BWD_X6_256_EXT: return launch_bwd<256, 19, NEFES_XYZ_EXTERNAL32, 6>(a, st);
        case BWD_X6_128: return launch_bwd<128, 131, NEFES_XYZ_FREQ10, 6>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 2
int nefes_bwd_launch_part2(int which, const FieldBwdArgs& a, hipStream_t st) {
    switch (which) {
        case BWD_STATIC_256: return launch_bwd<256, 19, NEFES_XYZ_FREQ10, 0, false>(a, st);
        case BWD_STATIC_128: return launch_bwd<128, 131, NEFES_XYZ_FREQ10, 0, false>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 3
int nefes_bwd_launch_part3(int which, const FieldBwdArgs& a, hipStream_t st) {
    switch (which) {
        case BWD_TRAIN_256_FULL: return launch_bwd<256, 19, NEFES_XYZ_FREQ10, 6, true, true>(a, st);
        case BWD_TRAIN_256_STATIC: return launch_bwd<256, 19, NEFES_XYZ_FREQ10, 0, false, true>(a, st);
        case BWD_TRAIN_128_FULL: return launch_bwd<128, 131, NEFES_XYZ_FREQ10, 0, true, true>(a, st);
        case BWD_TRAIN_128_STATIC: return launch_bwd<128, 131, NEFES_XYZ_FREQ10, 0, false, true>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#else   // part 0

static int field_bwd_impl(int x6, bool full, float* dacts, const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                          const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                          const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                          float* g_xyz_enc, float* g_viewdirs_s, void* stream) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    if (ext ? !g_xyz_enc : (!g_pts || (!pts && !(rays_o && rays_d && z)))) return NEFES_E_BADARG;
    if (full && !desc->has_transient) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[!full ? NEFES_STREAM_BWD_STATIC : (x6 ? NEFES_STREAM_BWD_FULL_X6 : NEFES_STREAM_BWD_FULL)];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldBwdArgs a;
    a.stream = (const char*)packed + si.slab_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_enc = g_xyz_enc; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + (full ? 6 : 1);
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    a.dacts = dacts;
    a.rows = nefes_train_row(desc->width, desc->feat_dim, NEFES_TB_END);
    hipStream_t st = (hipStream_t)stream;
    if (dacts) {   // train instances (frequency embedding): fused dX chain that also stores every layer's gradient vector
        if (ext) return NEFES_E_UNSUPPORTED;
        if (desc->width == 256 && desc->feat_dim == 16) {
            if (full) return nefes_bwd_launch_part3(BWD_TRAIN_256_FULL, a, st);
            return nefes_bwd_launch_part3(BWD_TRAIN_256_STATIC, a, st);
        }
        if (desc->width == 128 && desc->feat_dim == 128) {
            if (full) return nefes_bwd_launch_part3(BWD_TRAIN_128_FULL, a, st);
            return nefes_bwd_launch_part3(BWD_TRAIN_128_STATIC, a, st);
        }
        return NEFES_E_UNSUPPORTED;
    }
    if (!full) {   // static head only (fp32-MFMA instances)
        if (desc->width == 256 && desc->feat_dim == 16 && !ext) return nefes_bwd_launch_part2(BWD_STATIC_256, a, st);
        if (desc->width == 128 && desc->feat_dim == 128 && !ext) return nefes_bwd_launch_part2(BWD_STATIC_128, a, st);
        return NEFES_E_UNSUPPORTED;
    }
    if (x6) {
        if (desc->width == 256 && desc->feat_dim == 16 && !ext) return nefes_bwd_launch_part1(BWD_X6_256, a, st);
        if (desc->width == 256 && desc->feat_dim == 16 && ext) return nefes_bwd_launch_part1(BWD_X6_256_EXT, a, st);
        if (desc->width == 128 && desc->feat_dim == 128 && !ext) return nefes_bwd_launch_part1(BWD_X6_128, a, st);
        return NEFES_E_UNSUPPORTED;
    }
    if (desc->width == 256 && desc->feat_dim == 16 && !ext) return launch_bwd<256, 19, NEFES_XYZ_FREQ10>(a, st);
    if (desc->width == 128 && desc->feat_dim == 128 && !ext) return launch_bwd<128, 131, NEFES_XYZ_FREQ10>(a, st);
    if (desc->width == 256 && desc->feat_dim == 16 && ext) return launch_bwd<256, 19, NEFES_XYZ_EXTERNAL32>(a, st);
    return NEFES_E_UNSUPPORTED;
}

extern "C" int nefes_field_bwd(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                               const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                               const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                               float* g_xyz_enc, float* g_viewdirs_s, void* stream) {
    return field_bwd_impl(0, true, nullptr, desc, packed, N, S, rays_o, rays_d, z, pts, viewdirs, raw_t, g_raw_t, masks, g_pts, g_xyz_enc,
                          g_viewdirs_s, stream);
}

extern "C" int nefes_field_bwd_x6(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                  const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                  float* g_xyz_enc, float* g_viewdirs_s, void* stream) {
    return field_bwd_impl(6, true, nullptr, desc, packed, N, S, rays_o, rays_d, z, pts, viewdirs, raw_t, g_raw_t, masks, g_pts, g_xyz_enc,
                          g_viewdirs_s, stream);
}

extern "C" int nefes_field_bwd_static(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                      const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                      const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                      float* g_viewdirs_s, void* stream) {
    return field_bwd_impl(0, false, nullptr, desc, packed, N, S, rays_o, rays_d, z, pts, viewdirs, raw_t, g_raw_t, masks, g_pts, nullptr,
                          g_viewdirs_s, stream);
}

extern "C" int nefes_field_bwd_train(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                     const float* rays_d, const float* z, const float* viewdirs, const float* raw_t,
                                     const float* g_raw_t, const uint32_t* masks, float* dacts, float* g_pts,
                                     float* g_viewdirs_s, void* stream) {
    if (!dacts || (mode != NEFES_FIELD_STATIC && mode != NEFES_FIELD_FULL)) return NEFES_E_BADARG;
    const bool full = mode == NEFES_FIELD_FULL;
    // the FULL instance at width 256 runs on the bf16x6 stream, everything else on the fp32 streams
    const int x6 = (full && desc && desc->width == 256 && desc->feat_dim == 16) ? 6 : 0;
    return field_bwd_impl(x6, full, dacts, desc, packed, N, S, rays_o, rays_d, z, nullptr, viewdirs, raw_t, g_raw_t, masks, g_pts, nullptr,
                          g_viewdirs_s, stream);
}
#endif   // NEFES_TU_PART
