// Fused field forward: pts = o + d*z -> frequency embedding -> 8-layer skip MLP -> heads, one launch.
// Replaces run_network_NeRFH_NFF + Embedder.embed + NeRFH_NFF.forward
// (script/models/nerfh_nff.py:168-231, :234-270, :525-576) and rendering.py:114,142 (pts).
//
// Roofline: MFMA-bound (fp32 32x32x2 at 157.3 TFLOP/s); algorithmic work per sample = 2 * MACs of
// SURVEY.md §8d (491 264 sigma-only / 665 088 full at Wd=256,C=16).  Weights: LDS-DMA ring, read once
// per 128-sample workgroup tile from L2 (2.7 MB stream, resident).  Activations never leave registers.
#define NEFES_SLAB_KIB NEFES_FWD_SLAB_KIB
// every run of compiler-placed MFMAs ends with field_common.h mfma_results_fence: hipcc pads an MFMA's result against its own vector
// instructions, but takes the first path it finds to the MFMA where two join (seen: 7 of 18 wait states on the path that skips the
// mask stores) and does not look into the asm statements of the operand functors at all (tools/hazard_lint.py rules B1 / B2)
#define NEFES_ASM_READS_ACC
#include "field_common.h"
#include "../../include/nefes_hip.h"

struct FieldFwdArgs {
    const char* stream;      // weight slabs
    const float* bias;       // bias block (floats)
    uint32_t n_slabs, bias_floats;
    const float* rays_o;     // [N,3] or null
    const float* rays_d;
    const float* z;          // [N,S]
    const float* pts;        // [M,3] or null
    const float* xyz_enc;    // [M,32] (NEFES_XYZ_EXTERNAL32) or null
    const float* viewdirs;   // [N,3]
    float* raw_t;            // [N][R][S]
    uint32_t* masks;         // [tiles32][MW][64] or null
    int N, S, R, C;
    long long M;
    int n_tiles;             // 128-sample workgroup tiles
    float* acts;             // TRAIN instances: [n_tiles][rows][128] pre-activations + embeddings (layout.h row map)
    int rows;
};

// TRAIN: dump a layer's accumulators (pre-activation, bias included) as rows [row0, row0 + 32 NT) of this tile.
// One accumulator register = two 128-byte row segments (lane halves hold rows rho and rho + 4): full-rate stores.
template <int NT>
__device__ __forceinline__ void train_save(float* tile_base, uint32_t voff, int row0, const f32x16 (&X)[NT]) {
    float* p = tile_base + (size_t)(row0 >> 5) * 4096 + voff;      // layout.h nefes_train_off: voff = nefes_train_lane_off
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[t * 4096 + nefes_rho(0, r) * 16] = X[t][r];
}

// MODE: NEFES_FIELD_SIGMA / STATIC / FULL.  W: MLP width.  NTR: tiles of the rgb+feature head.
// ENC: NEFES_XYZ_FREQ10 (embedding computed here) / NEFES_XYZ_EXTERNAL32 (32 features per sample read from xyz_enc).
// TRAIN: additionally write every hidden layer's pre-activation and both embeddings to a.acts (weight-gradient pass).
template <int W, int NTR, int MODE, int ENC, bool TRAIN = false>
__global__ __launch_bounds__(256, 1) void field_fwd_kernel(FieldFwdArgs a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;   // tiles / k-steps
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128);
    constexpr int ES = ENC == NEFES_XYZ_EXTERNAL32 ? NEFES_X_STEPS : NEFES_E_STEPS;   // k-steps of the xyz embedding
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring_base = smem;
    float* bias_lds = (float*)(smem + NEFES_RING_SLOTS * NEFES_SLAB_BYTES);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    for (uint32_t i = threadIdx.x; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
    WeightRing<NEFES_RING_SLOTS> ring;
    ring.init(a.stream, a.n_slabs, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_base, wave, lane);
    __syncthreads();
    const char* ring_lane = ring_base + lane * 16;
    const char* bias_half = (const char*)bias_lds + 16 * h;
    ring.prime(ring_lane);

    // bias block offsets (floats), in stream order: L1..L8, SIG, FINAL, DIR, RGB, T0, T1, T2, TH
    constexpr int B_SIG = 8 * W, B_FINAL = B_SIG + 32, B_DIR = B_FINAL + W, B_RGB = B_DIR + W / 2,
                  B_T0 = B_RGB + 32 * NTR, B_T1 = B_T0 + W / 2, B_T2 = B_T1 + W / 2, B_TH = B_T2 + W / 2;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long m_raw = (long long)tile * 128 + wave * 32 + j;
        const bool valid = m_raw < a.M;
        const long long m = valid ? m_raw : a.M - 1;
        const int ray = (int)(m / a.S);
        const int smp = (int)(m - (long long)ray * a.S);
        // ---- the only global loads of the tile: raw inputs, completed by loads_landed() before any use ----
        float in_o[3] = {0.f, 0.f, 0.f}, in_d[3] = {0.f, 0.f, 0.f}, in_z = 0.f, v[3] = {0.f, 0.f, 0.f};
        float E[ES];
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
#pragma unroll
            for (int s = 0; s < ES; ++s) E[s] = a.xyz_enc[m * 32 + 2 * s + h];    // compact slots: feature 2s+h
        } else if (a.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) in_o[c] = a.pts[m * 3 + c];
        } else {
            in_z = a.z[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) { in_o[c] = a.rays_o[ray * 3 + c]; in_d[c] = a.rays_d[ray * 3 + c]; }
        }
        if (MODE != NEFES_FIELD_SIGMA) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[ray * 3 + c];
        }
        loads_landed();
        pin(in_o); pin(in_d); pin(in_z); pin(v);
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            pin(E);
        } else {
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = a.pts ? in_o[c] : add_rn(in_o[c], mul_rn(in_d[c], in_z));   // rendering.py:114,142
            embed_slots<NEFES_N_FREQ_XYZ>(E, x, h);
        }
        float Dv[NEFES_D_STEPS];
        if (MODE != NEFES_FIELD_SIGMA) embed_slots<NEFES_N_FREQ_DIR>(Dv, v, h);
        // ReLU masks are lane-private words of whole 32-sample tiles (the buffer is padded to whole tiles), so they are
        // written unconditionally: no per-lane predicate on the store path.
        // Mask words: wave-uniform tile base (SGPRs) + a uniform running word counter; the only per-lane part is `lane`.
        // Words are emitted in layout order (layers 1..8, dir, transient 0,2,4).  Lane-private words of whole 32-sample
        // tiles (the buffer is padded to whole tiles) => stored unconditionally, no per-lane predicate.
        uint32_t* mask_tile = (MODE != NEFES_FIELD_SIGMA && a.masks)
                                  ? a.masks + ((size_t)((long long)tile * 4 + wave) * MW) * 64
                                  : nullptr;
        int mask_word = 0;
        constexpr int WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;
        auto put_masks = [&](const uint32_t* bits, int n) {
            if (mask_tile) {
                for (int w = 0; w < n; ++w) mask_tile[(mask_word + w) * 64 + lane] = bits[w];
                mask_word += n;
            }
        };
        // per-lane base of this sample's column in raw_t; channel c lives at + c*S.  Re-materialised (pinned) right before
        // each group of stores so the compiler does not keep 25 precomputed 64-bit addresses alive across the MLP.
        const size_t raw_off = (size_t)ray * a.R * a.S + smp;
        auto raw_col = [&]() {
            float* pcol = a.raw_t + raw_off;
            asm volatile("" : "+v"(pcol));
            return pcol;
        };
        auto bias_at = [&](int off_floats) { return BiasInit{bias_half + off_floats * 4}; };
        const ArrayIn<ES> in_E{E};
        const ArrayIn<NEFES_D_STEPS> in_D{Dv};
        float* act_tile = nullptr;                                    // wave-uniform
        const uint32_t act_voff = nefes_train_lane_off(wave, j, h);
        if constexpr (TRAIN) {
            act_tile = a.acts + (size_t)tile * a.rows * 128;
            const uint32_t emb_off = (uint32_t)(((wave * 32 + j) >> 4) * 512 + h * 16 + (j & 15));   // row 2s+h of an embedding block
            float* pe = act_tile + (size_t)(nefes_train_row(W, 0, NEFES_TB_E) >> 5) * 4096 + emb_off;
#pragma unroll
            for (int s = 0; s < ES; ++s) pe[(s >> 4) * 4096 + 2 * (s & 15) * 16] = E[s];     // slot (s,h) -> row 2s+h
            if (MODE != NEFES_FIELD_SIGMA) {
                float* pd = act_tile + (size_t)(nefes_train_row(W, 0, NEFES_TB_DV) >> 5) * 4096 + emb_off;
#pragma unroll
                for (int s = 0; s < NEFES_D_STEPS; ++s) pd[2 * s * 16] = Dv[s];
            }
        }
        auto save_trunk = [&](int layer, const f32x16 (&X)[NTW]) {   // layer 1..9 (9 = xyz_encoding_final)
            if constexpr (TRAIN) train_save<NTW>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_L1) + (layer - 1) * W, X);
        };
        auto save_half = [&](int block, const f32x16 (&X)[NTH]) {
            if constexpr (TRAIN) train_save<NTH>(act_tile, act_voff, nefes_train_row(W, 0, block), X);
        };

        // Ping-pong accumulators: a layer reads its input straight out of the other array (consumer-side ReLU).
        f32x16 A[NTW], B[NTW];
        uint32_t bits[WT];
        auto clear_bits = [&]() {
#pragma unroll
            for (int w = 0; w < WT; ++w) bits[w] = 0u;
        };
        auto sigma_head = [&](const f32x16 (&X)[NTW]) {
            // static_sigma on h8 = relu(X) (nerfh_nff.py:485,555): one tile, row 0 = (half 0, register 0)
            f32x16 sg[1];
            mma_run<1, HS, 0, true>(ring, ring_lane, ReluIn<NTW>{X}, bias_at(B_SIG), sg);
            if (valid && h == 0) {
                const int ch = (MODE == NEFES_FIELD_SIGMA) ? 0 : 3 + a.C;
                raw_col()[(size_t)ch * a.S] = softplus_ref(sg[0][0]);
            }
        };
        // ---- layer 1: 63 -> W (its bias rides on the first k-step as the C operand) ----
        mma_run<NTW, ES, 0, true>(ring, ring_lane, in_E, bias_at(0), A);
        save_trunk(1, A);
        // ---- layers 2..8 (+ xyz_encoding_final as layer 9 in STATIC/FULL), two per iteration: A -> B -> A ----
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            clear_bits();
            // masks are recorded only when a backward pass can follow (STATIC, FULL); the sigma-only pass saves the VALU op
            if constexpr (MODE != NEFES_FIELD_SIGMA)
                mma_run<NTW, HS, 0, true>(ring, ring_lane, ReluCapture<NTW, WT>{A, bits}, bias_at((l1 - 1) * W), B);
            else
                mma_run<NTW, HS, 0, true>(ring, ring_lane, ReluIn<NTW>{A}, bias_at((l1 - 1) * W), B);
            put_masks(bits, WT);                                      // mask of layer l1-1 (the producer of A)
            save_trunk(l1, B);
            if (p == 3) {
                if (MODE == NEFES_FIELD_SIGMA) break;                 // layer 8 is the last; B holds its pre-activation
                sigma_head(B);
            }
            clear_bits();
            if constexpr (MODE != NEFES_FIELD_SIGMA)
                mma_run<NTW, HS, 0, true>(ring, ring_lane, ReluCapture<NTW, WT>{B, bits},
                                          bias_at(l2 <= 8 ? (l2 - 1) * W : B_FINAL), A);
            else
                mma_run<NTW, HS, 0, true>(ring, ring_lane, ReluIn<NTW>{B}, bias_at(l2 <= 8 ? (l2 - 1) * W : B_FINAL), A);
            if (p == 1) mma_run<NTW, ES, 0, false>(ring, ring_lane, in_E, ZeroInit{}, A);   // skip: + W5[:, :63] e
            put_masks(bits, WT);                                      // mask of layer l1 (the producer of B)
            save_trunk(l2, A);
        }
        if constexpr (MODE == NEFES_FIELD_SIGMA) sigma_head(B);
        if constexpr (MODE != NEFES_FIELD_SIGMA) {
            // A holds xyz_encoding_final (no activation, :559)
            f32x16 acc2[NTH], acc3[NTH];
            uint32_t bits2[WH];
            auto clear2 = [&]() {
#pragma unroll
                for (int w = 0; w < WH; ++w) bits2[w] = 0u;
            };
            // ---- dir_encoding: cat[final, dir-emb] -> W/2 ----
            mma_run<NTH, HS, 0, true>(ring, ring_lane, IdentIn<NTW>{A}, bias_at(B_DIR), acc2);
            mma_run<NTH, NEFES_D_STEPS, 0, false>(ring, ring_lane, in_D, ZeroInit{}, acc2);
            save_half(NEFES_TB_DIR, acc2);
            // ---- static_rgb on relu(dir): W/2 -> 3+C, no activation (:487-490) ----
            {
                f32x16 ar[NTR];
                clear2();
                mma_run<NTR, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_RGB), ar);
                put_masks(bits2, WH);                                 // dir_encoding
                if (valid) {
                    // channel of (tile t, register r, half h) = 32t + rho(0,r) + 4h: per-lane part folded into the base,
                    // the rest is a wave-uniform offset
                    float* ph = raw_col() + (size_t)(4 * h) * a.S;
#pragma unroll
                    for (int t = 0; t < NTR; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int cu = 32 * t + nefes_rho(0, r);
                            if (cu + 4 * h < 3 + a.C) ph[(size_t)cu * a.S] = ar[t][r];
                        }
                }
            }
            if constexpr (MODE == NEFES_FIELD_FULL) {
                // ---- transient_encoding.{0,2,4} ----
                mma_run<NTH, HS, 0, true>(ring, ring_lane, IdentIn<NTW>{A}, bias_at(B_T0), acc2);
                mma_run<NTH, NEFES_D_STEPS, 0, false>(ring, ring_lane, in_D, ZeroInit{}, acc2);
                save_half(NEFES_TB_T0, acc2);
                clear2();
                mma_run<NTH, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_T1), acc3);
                put_masks(bits2, WH);                                 // transient_encoding.0
                save_half(NEFES_TB_T1, acc3);
                clear2();
                mma_run<NTH, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc3, bits2}, bias_at(B_T2), acc2);
                put_masks(bits2, WH);                                 // transient_encoding.2
                save_half(NEFES_TB_T2, acc2);
                // ---- transient heads: rows 0..2 rgb (sigmoid), 3 sigma (softplus), 4 beta (softplus) ----
                f32x16 th[1];
                clear2();
                mma_run<1, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_TH), th);
                put_masks(bits2, WH);                                 // transient_encoding.4
                if (valid) {
                    float* o = raw_col() + (size_t)(3 + a.C + 1) * a.S;
                    if (h == 0) {
                        o[0] = sigmoid_ref(th[0][0]);
                        o[(size_t)a.S] = sigmoid_ref(th[0][1]);
                        o[(size_t)2 * a.S] = sigmoid_ref(th[0][2]);
                        o[(size_t)3 * a.S] = softplus_ref(th[0][3]);
                    } else {
                        o[(size_t)4 * a.S] = softplus_ref(th[0][0]);
                    }
                }
            }
        }
    }
    ring.drain();
}

template <int W, int NTR, int MODE, int ENC, bool TRAIN = false>
static int launch_fwd(const FieldFwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_RING_SLOTS * NEFES_SLAB_BYTES + ((a.bias_floats * 4 + 255) / 256) * 256;
    auto k = field_fwd_kernel<W, NTR, MODE, ENC, TRAIN>;
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

extern "C" size_t nefes_field_mask_bytes(const NefesNetDesc* desc, int64_t M) {
    if (!desc || M <= 0) return 0;
    const int64_t tiles32 = ((M + 127) / 128) * 4;
    return (size_t)tiles32 * nefes_mask_words(desc->width) * 64 * 4;
}

static int field_fwd_impl(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                          const float* rays_d, const float* z, const float* pts, const float* xyz_enc,
                          const float* viewdirs, float* raw_t, uint32_t* masks, float* acts, void* stream) {
    if (!desc || !packed || !raw_t || N <= 0 || S <= 0) return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    if (ext ? !xyz_enc : (!pts && !(rays_o && rays_d && z))) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_SIGMA && !viewdirs) return NEFES_E_BADARG;
    if (mode == NEFES_FIELD_FULL && !desc->has_transient) return NEFES_E_BADARG;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const int sk = mode == NEFES_FIELD_SIGMA ? NEFES_STREAM_FWD_SIGMA
                 : mode == NEFES_FIELD_STATIC ? NEFES_STREAM_FWD_STATIC : NEFES_STREAM_FWD_FULL;
    const NefesStreamInfo& si = info.stream[sk];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldFwdArgs a;
    a.stream = (const char*)packed + si.slab_off;
    a.bias = (const float*)((const char*)packed + si.bias_off);
    a.n_slabs = si.n_slabs; a.bias_floats = si.bias_floats;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.xyz_enc = xyz_enc; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.masks = masks; a.acts = acts;
    a.rows = nefes_train_row(desc->width, desc->feat_dim, NEFES_TB_END);
    a.N = N; a.S = S; a.C = desc->feat_dim;
    a.R = mode == NEFES_FIELD_SIGMA ? 1 : (mode == NEFES_FIELD_STATIC ? 3 + a.C + 1 : 3 + a.C + 6);
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    const int W = desc->width, C = desc->feat_dim;
    const int ntr = (3 + C + 31) / 32;
    if (acts) {   // train-mode instances: frequency embedding, heads present
        if (mode == NEFES_FIELD_SIGMA || desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
#define NEFES_DISPATCH_TRAIN(WW, NN)                                                                            \
    if (W == WW && ntr == NN) {                                                                                 \
        if (mode == NEFES_FIELD_STATIC) return launch_fwd<WW, NN, NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, true>(a, st); \
        return launch_fwd<WW, NN, NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, true>(a, st);                             \
    }
        NEFES_DISPATCH_TRAIN(256, 1)
        NEFES_DISPATCH_TRAIN(128, 5)
#undef NEFES_DISPATCH_TRAIN
        return NEFES_E_UNSUPPORTED;
    }
#define NEFES_DISPATCH(WW, NN, EE)                                                              \
    if (W == WW && ntr == NN && desc->xyz_encoding == EE) {                                         \
        if (mode == NEFES_FIELD_SIGMA) return launch_fwd<WW, NN, NEFES_FIELD_SIGMA, EE>(a, st);   \
        if (mode == NEFES_FIELD_STATIC) return launch_fwd<WW, NN, NEFES_FIELD_STATIC, EE>(a, st); \
        return launch_fwd<WW, NN, NEFES_FIELD_FULL, EE>(a, st);                                   \
    }
    NEFES_DISPATCH(256, 1, NEFES_XYZ_FREQ10)       /* BASELINE metric shape: Wd=256, C=16 */
    NEFES_DISPATCH(128, 5, NEFES_XYZ_FREQ10)       /* reference defaults:    Wd=128, C=128 */
    NEFES_DISPATCH(256, 1, NEFES_XYZ_EXTERNAL32)   /* BASELINE config 4: hash-grid embedding in front of the same MLP */
#undef NEFES_DISPATCH
    return NEFES_E_UNSUPPORTED;
}

extern "C" int nefes_field_fwd(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                               const float* rays_d, const float* z, const float* pts, const float* xyz_enc,
                               const float* viewdirs, float* raw_t, uint32_t* masks, void* stream) {
    return field_fwd_impl(desc, packed, mode, N, S, rays_o, rays_d, z, pts, xyz_enc, viewdirs, raw_t, masks, nullptr, stream);
}

extern "C" size_t nefes_train_rows(const NefesNetDesc* desc) {
    return desc ? (size_t)nefes_train_row(desc->width, desc->feat_dim, NEFES_TB_END) : 0;
}

extern "C" int nefes_train_row_offset(const NefesNetDesc* desc, int block) {
    if (!desc || block < 0 || block > NEFES_TB_END) return NEFES_E_BADARG;
    return nefes_train_row(desc->width, desc->feat_dim, block);
}

extern "C" int nefes_field_fwd_train(const NefesNetDesc* desc, const void* packed, int mode, int N, int S,
                                     const float* rays_o, const float* rays_d, const float* z, const float* pts,
                                     const float* viewdirs, float* raw_t, float* acts, uint32_t* masks, void* stream) {
    if (!acts) return NEFES_E_BADARG;
    return field_fwd_impl(desc, packed, mode, N, S, rays_o, rays_d, z, pts, nullptr, viewdirs, raw_t, masks, acts, stream);
}
