// Fused field forward with every product as an fp16 two-part split product (field_h3.h): sigma-only coarse pass and the full
// fine pass.  Same function as field_fwd_kernel / field_fwd_x6_kernel (script/models/nerfh_nff.py:192-231,525-576 +
// rendering.py:114,142): pts = o + d*z -> frequency embedding (or a supplied 32-feature encoding) -> 8-layer skip MLP -> heads,
// same raw_t layout, same ReLU-mask words.  Every product runs on v_mfma_f32_32x32x16_f16 as hh + hl + lh of power-of-two
// scaled (hi, lo) fp16 pairs: half the matrix-core work of the bf16x6 kernels at fp32-level accuracy.
// Weight stream: 32 KiB slabs = 16 units of two 1 KiB groups (hi, lo), moved through registers into a two-slot LDS ring
// (field_h3.h StagedRing).  The xyz embedding (32 slots per lane) waits
// in LDS between layer 1 and the skip at layer 5 instead of in registers: with one accumulator set in VGPRs (the compiler keeps
// the set the vector ALU reads there) the kernel has no 32 registers to spare.
// (layout.h is included below; the two sizes are repeated there as NEFES_H3_FWD_SLAB_KIB / _128 and checked against these)
#if defined(NEFES_TU_PART) && NEFES_TU_PART >= 2 && NEFES_TU_PART % 2 == 0
#define NEFES_TU_W128          // even parts from 2 on hold the Wd = 128 instances
#define NEFES_SLAB_KIB 16      // 2 x 16 KiB of ring, two workgroups per CU (see launch_h3)
#else
#define NEFES_SLAB_KIB 32
#endif
#include <stdlib.h>
// Accumulators read by asm statements: the Wd = 128 objects (-amdgpu-mfma-vgpr-form: the functors' asm reads the MFMAs' own VGPRs) and
// the Wd = 256 objects (H3_ACC_READ_ASM below): compiler-placed runs end with field_common.h mfma_results_fence
#if defined(NEFES_TU_W128) || !defined(H3_NO_ACC_READ_ASM)
#define NEFES_ASM_READS_ACC
#define NEFES_FENCE_TILES
#endif
#if defined(NEFES_TU_W128)
#define NEFES_ACC_VGPR_FORM
#endif

#ifndef NEFES_FWD_CONSUMER_BIAS
#define NEFES_FWD_CONSUMER_BIAS 1   /* 0: bias tiles written by the producing run (A/B builds) */
#endif
#include "field_common.h"
#include "../../include/nefes_hip.h"

#include "field_x6.h"
#include "hashgrid.h"
#define NEFES_XYZ_HASHGRID_FUSED 2   /* kernel-internal ENC value: a NEFES_XYZ_EXTERNAL32 network whose 32 features the kernel gathers itself */
#if !defined(NEFES_TU_W128) && !defined(H3_NO_ACC_READ_ASM)
#define H3_ACC_READ_ASM        // Wd = 256 objects: source tiles are read out of their AGPRs inside the MFMA gaps (field_h3.h acc_read)
#endif
#include "field_h3.h"
#define NEFES_H3_SLOTS 2   // 64 KiB ring (StagedRing: two slots)

struct FieldFwdH3Args {
    const char* stream;
    const float* bias;       // bias blocks followed by the segment exponent table (NefesStreamInfo.scale_off)
    uint32_t n_slabs, bias_floats, scale_off;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    const float* xyz_enc;    // [M,32] (NEFES_XYZ_EXTERNAL32) or null
    const float* viewdirs;   // [N,3] (FULL)
    float* raw_t;            // [N][R][S]
    uint32_t* masks;         // [tiles32][MW][64] or null (FULL)
    int N, S, R, C;
    long long M;
    int n_tiles;
    uint32_t s_magic, s_shift;   // m / S == mulhi(m, s_magic) >> s_shift for every m < 2^31 (host: magic_div)
    float* acts;             // TRAIN instances: [n_tiles][rows][128] pre-activations + embeddings (layout.h row map)
    int rows;
    int z_row;               // 1: `z` is ONE row of S depths shared by every ray (scalar near / far, no jitter: rendering.py:96-100)
    const float2* hg_table;  // NEFES_XYZ_HASHGRID_FUSED: the hash-grid table and its level geometry (hashgrid.h)
    HgGeom hg;
    int gout;                // FH instances: W / 2 = the channels of relu(dir_encoding) written in place of the feature head's outputs
};

// TRAIN: tiles X[T0 .. T0+NT) hold a layer's pre-activations times 2^es; rows [row0, row0 + 32 NT) of this tile of `acts` get
// the true values (inv = 2^-es).  One register = two 128-byte row segments (lane halves hold rows rho and rho + 4).
// (Round 6 tried the register PAIRS swapped across 16-lane rows so that every store instruction writes two full 128-byte lines --
// in isolation full-line non-temporal writes run at 5.2-6.0 TB/s against 3.0-3.4 for these half lines, tools/probe/write_probe.hip --
// and the kernel did not move: 3.465 vs 3.47 ms.  What the stores cost here, 0.95 of 3.47 ms by the -DNEFES_TRAIN_NO_STORES build, is
// the wave's own ring loads waiting behind them in the in-order vmcnt, not the write pattern: DESIGN.md section 7 item 2.)
template <int NT, int T0, int NX>
__device__ __forceinline__ void train_save_h3(float* tile_base, uint32_t voff, int row0, const f32x16 (&X)[NX], float inv) {
    float* p = tile_base + (size_t)(row0 >> 5) * 4096 + voff;      // layout.h nefes_train_off: voff = nefes_train_lane_off
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifndef NEFES_TRAIN_NO_STORES      /* (ablation builds: what the stores cost, tools/ab_side.sh) */
            __builtin_nontemporal_store(X[T0 + t][r] * inv, &p[t * 4096 + nefes_rho(0, r) * 16]);
#endif
        }
}

// MODE: NEFES_FIELD_SIGMA, NEFES_FIELD_STATIC (static head only: the TRAIN instances of a coarse network) or NEFES_FIELD_FULL;
// ENC: NEFES_XYZ_FREQ10 or NEFES_XYZ_EXTERNAL32 (hash grid); W = 128 or 256; NTR = tiles of the rgb+feature head = the head class
// of layout.h (1: 3 + C <= 32, e.g. BASELINE's C = 16; 5: 3 + C <= 144, e.g. the reference's FEATURE_DIM = 128) -- C itself is a
// run-time argument.  TRAIN: every hidden layer's pre-activation and both embeddings also go to a.acts (weight gradients).
// FH ("factored head", round 5): the network's static rgb+feature head is LINEAR in g = relu(dir_encoding) and so is compositing
// (raw2outputs: feat = sum_s w_s (W_f g_s + b_f), nerfh_nff.py:119-125 -- no activation on the head when C > 0, :487-490), so a network
// whose head has more channels than g (the reference's default: 3 + 128 against 64) emits g itself -- W/2 channels + a channel of ones
// (for b_f sum_s w_s) in the feature channels' place, the 3 colour channels from a 3-row head (head class 0) -- and the caller applies
// W_f once per RAY to the composited g (nefes_amd/render.py).  44 of the head's 60 MFMAs per 32 samples and 63 of 137 raw channels go.
template <int MODE, int ENC, int W = 256, int NTR = 1, bool TRAIN = false, bool FH = false>
__global__ __launch_bounds__(256, W == 128 ? 2 : 1) void field_fwd_h3_kernel(FieldFwdH3Args a) {
    static_assert(NEFES_SLAB_KIB == (W == 128 ? NEFES_H3_FWD_SLAB_KIB_128 : NEFES_H3_FWD_SLAB_KIB), "ring slab size != the packer's for this width");
    constexpr int NTW = W / 32, NTH = W / 64;
    constexpr int ES = ENC != NEFES_XYZ_FREQ10 ? NEFES_X_STEPS : NEFES_E_STEPS;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128), WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;
    constexpr int NSEG = MODE == NEFES_FIELD_SIGMA ? NEFES_H3F_SIG + 1 : (MODE == NEFES_FIELD_STATIC ? NEFES_H3F_N_STATIC : NEFES_H3F_N);   // segments of the stream
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring_base = smem;
    float* bias_lds = (float*)(smem + NEFES_H3_SLOTS * NEFES_SLAB_BYTES);
    // embedding stash: [wave][ES slots][64 lanes] floats behind the bias block (launch_h3 sizes the allocation)
    float* e_lds = bias_lds + ((a.bias_floats + 63) / 64) * 64 + (threadIdx.x >> 6) * (ES * 64) + (threadIdx.x & 63);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    for (uint32_t i = threadIdx.x; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
    // NEFES_XYZ_HASHGRID_FUSED: the level geometry waits in LDS behind the embedding stash (read out of the kernel arguments at its
    // uses, hipcc hoists the eighty scalar loads out of the tile loop and spills them); published by ring.prime()'s barrier
    HgGeom* hg_lds = (HgGeom*)(bias_lds + ((a.bias_floats + 63) / 64) * 64 + 4 * ES * 64);
    if constexpr (ENC == NEFES_XYZ_HASHGRID_FUSED) {
        if (threadIdx.x == 0) *hg_lds = a.hg;
    }
    StagedRing ring;
    ring.init(a.stream, a.n_slabs, ring_base, wave, lane);
    const char* ring_lane = ring_base + lane * 16;
    const char* bias_half = (const char*)bias_lds + 16 * h;
    // scale table (layout.h): per segment (weight exponent, row bound), then max |b| per bias block
    const int* tab_i = (const int*)(bias_lds + a.scale_off);
    const float* tab_f = bias_lds + a.scale_off;
    auto wexp = [&](int seg) { return tab_i[nefes_h3_tab_exp(seg)]; };
    auto rowb = [&](int seg) { return tab_f[nefes_h3_tab_bound(seg)]; };
    auto bmax = [&](int blk) { return tab_f[nefes_h3_tab_bias(NSEG, blk)]; };
    ring.prime(ring_lane);                                       // (its barrier also publishes the bias block)
    // bias block offsets (floats), in stream order: L1..L8, SIG, FINAL, DIR, RGB, T0, T1, T2, TH (as in field_fwd.hip)
    constexpr int B_SIG = 8 * W, B_FINAL = B_SIG + 32, B_DIR = B_FINAL + W, B_RGB = B_DIR + W / 2,
                  B_T0 = B_RGB + 32 * NTR, B_T1 = B_T0 + W / 2, B_T2 = B_T1 + W / 2, B_TH = B_T2 + W / 2;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        // sample / ray indices are recomputed where they are needed (loads here, three output points below) rather than kept
        // alive across the tile: every long-lived per-lane value costs a register the accumulators need (M < 2^31: host check)
        auto locate = [&](uint32_t& m, uint32_t& ray, uint32_t& smp) {
            const uint32_t m_raw = (uint32_t)tile * 128u + (uint32_t)(wave * 32 + j);
            const bool ok = m_raw < (uint32_t)a.M;
            m = ok ? m_raw : (uint32_t)a.M - 1u;
            ray = a.s_magic ? __umulhi(m, a.s_magic) >> a.s_shift : m;   // = m / S without a reciprocal kept in a register
            smp = m - ray * (uint32_t)a.S;
            return ok;
        };
        uint32_t m, ray, smp;
        locate(m, ray, smp);
        float in_o[3] = {0.f, 0.f, 0.f}, in_d[3] = {0.f, 0.f, 0.f}, in_z = 0.f;
        float E[ES];
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
#pragma unroll
            for (int s = 0; s < ES; ++s) E[s] = a.xyz_enc[(size_t)m * 32 + 2 * s + h];    // compact slots: feature 2s+h
        } else if (ENC != NEFES_XYZ_HASHGRID_FUSED && a.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) in_o[c] = a.pts[(size_t)m * 3 + c];
        } else {
            in_z = a.z[a.z_row ? smp : m];
#pragma unroll
            for (int c = 0; c < 3; ++c) { in_o[c] = a.rays_o[ray * 3 + c]; in_d[c] = a.rays_d[ray * 3 + c]; }
        }
        float mE;                                                   // largest embedding magnitude of the sample (true units)
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            mE = pair_max(array_max(E));
        } else if constexpr (ENC == NEFES_XYZ_HASHGRID_FUSED) {
            // the hash-grid encoding of pts = o + d z (rendering.py:114,142; nerfh_tcnn.py:151-156) gathered HERE: the [M, 32]
            // encoding (10 GB per fine pass of the 854x480 frame), the stand-alone gather launch and the torch expression for pts go
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = add_rn(in_o[c], mul_rn(in_d[c], in_z));
            hg_encode_slots(E, x, h, *hg_lds, a.hg_table);
            mE = pair_max(array_max(E));
        } else {
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = a.pts ? in_o[c] : add_rn(in_o[c], mul_rn(in_d[c], in_z));   // rendering.py:114,142
            embed_slots<NEFES_N_FREQ_XYZ>(E, x, h);
            mE = fmaxf(fmaxf(1.f, fabsf(x[0])), fmaxf(fabsf(x[1]), fabsf(x[2])));   // sin/cos <= 1, then x itself
        }
#pragma unroll
        for (int s = 0; s < ES; ++s) e_lds[s * 64] = E[s];          // own lane's column only: no barrier needed
        float* act_tile = nullptr;                                  // wave-uniform
        const uint32_t act_voff = nefes_train_lane_off(wave, j, h);
        const uint32_t emb_off = (uint32_t)(((wave * 32 + j) >> 4) * 512 + h * 16 + (j & 15));   // row 2s+h of an embedding block
        if constexpr (TRAIN) {
            act_tile = a.acts + (size_t)tile * a.rows * 128;
            float* pe = act_tile + (size_t)(nefes_train_row(W, 0, NEFES_TB_E) >> 5) * 4096 + emb_off;
#pragma unroll
            for (int s = 0; s < ES; ++s) __builtin_nontemporal_store(E[s], &pe[(s >> 4) * 4096 + 2 * (s & 15) * 16]);     // slot (s,h) -> row 2s+h
        }
        auto save_trunk = [&](int layer, const f32x16 (&X)[NTW], int es) {   // layer 1..9 (9 = xyz_encoding_final)
            if constexpr (TRAIN) train_save_h3<NTW, 0>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_L1) + (layer - 1) * W, X, pow2i(-es));
        };
        // this lane's column of the tile's mask words: a running pointer, so that a layer's words leave at constant offsets
        // (a word index kept as an integer cost three vector instructions of 64-bit address arithmetic per store, at every layer
        // boundary, where the matrix pipe idles)
        uint32_t* mask_ptr = (MODE != NEFES_FIELD_SIGMA && a.masks) ? a.masks + ((size_t)((long long)tile * 4 + wave) * MW) * 64 + lane : nullptr;
        auto put_masks = [&](const uint32_t* bits, int n) {
            if (MODE != NEFES_FIELD_SIGMA && mask_ptr) {
                for (int w = 0; w < n; ++w) __builtin_nontemporal_store(bits[w], &mask_ptr[w * 64]);   // written once, read once by the backward: keep it out of the way of the weight stream in L2
                mask_ptr += n * 64;
            }
        };
        // column of this lane's sample in raw_t (null for the padding lanes of the last tile)
        auto raw_col = [&]() -> float* {
            uint32_t mm, rr, ss;
            if (!locate(mm, rr, ss)) return nullptr;
            return a.raw_t + ((size_t)rr * a.R * a.S + ss);
        };
        auto bias_at = [&](int off_floats, int es) { return BiasInitScaled{bias_half + off_floats * 4, pow2i(es)}; };
        // CB (inference instances): the trunk layers 1..8 start their tiles from the constant 0 and their consumers add the bias
        // (field_h3.h ReluBiasSplitH) -- sixteen vector instructions less per tile and layer.  TRAIN instances store the tiles as
        // pre-activations and keep the bias in the producer.
        constexpr bool CB = !TRAIN && W == 256 && NEFES_FWD_CONSUMER_BIAS;     // (Wd = 128, block-per-pair runs: measured 1-2 % slower)
        float2 nbias;
        f32x16 A[NTW], B[NTW];
        uint32_t bits[WT];
        auto clear_bits = [&]() {
#pragma unroll
            for (int w = 0; w < WT; ++w) bits[w] = 0u;
        };
        constexpr bool CAP = MODE != NEFES_FIELD_SIGMA;
        // Scale bookkeeping (field_h3.h), all per lane and identical on the two lanes of a sample:
        //   es_x  exponent an accumulator set carries (acc = 2^es_x * true value)
        //   M_x   upper bound of the set's largest (ReLU'd / absolute) true value, from the packer's row bounds
        //   tau   exponent the operand is brought to (operand = true value * 2^tau, largest magnitude below 2^15): tau =
        //         pick_exp(M_x), the conversion multiplies the accumulator by 2^(tau - es_x); output exponent = tau + weight exponent
        // tau_of: capped so that the output scale 2^(tau + ew) stays a finite float
        auto tau_of = [&](float M, int ew) {
            const int t = pick_exp(M);
            return t < 100 - ew ? t : 100 - ew;
        };
        auto sigma_head = [&](const f32x16 (&X)[NTW], int es_x, int tau_x) {      // tau_x: exponent chosen for relu(X)
            f32x16 sg[1];
            float mdummy = 0.f;
            const int es = tau_x + wexp(NEFES_H3F_SIG);
            if constexpr (CB) {
                const char* bp8 = bias_half + 7 * W * 4;                                  // the tiles are layer 8's, without its bias
                nbias = ReluBiasSplitH<false, NTW, WT>::prime(bp8);
                mma_run_h3<1, W / 16, 0, true>(ring, ring_lane, ReluBiasSplitH<false, NTW, WT>{X, bits, pow2i(tau_x - es_x), mdummy, bp8, pow2i(es_x), nbias},
                                               bias_at(B_SIG, es), sg);
            } else
            mma_run_h3<1, W / 16, 0, true>(ring, ring_lane, ReluSplitH<false, NTW, WT>{X, bits, pow2i(tau_x - es_x), mdummy},
                                           bias_at(B_SIG, es), sg);
            float* col = raw_col();
            if (col && h == 0) {
                const int ch = (MODE == NEFES_FIELD_SIGMA) ? 0 : 3 + a.C;
                __builtin_nontemporal_store(softplus_ref(sg[0][0] * pow2i(-es)), &col[(size_t)ch * a.S]);
            }
        };
        int es_a, es_b = 0;
        float M;                                                    // bound of the latest layer's output (true units)
        {
            const int tau = tau_of(mE, wexp(NEFES_H3F_L1));
            es_a = tau + wexp(NEFES_H3F_L1);
            if constexpr (CB) mma_run_h3<NTW, ES / 8, 0, true>(ring, ring_lane, LdsSplitH{e_lds, pow2i(tau)}, ZeroInit{}, A);    // layer 1 (bias: consumer)
            else mma_run_h3<NTW, ES / 8, 0, true>(ring, ring_lane, LdsSplitH{e_lds, pow2i(tau)}, bias_at(0, es_a), A);            // layer 1
            M = rowb(NEFES_H3F_L1) * mE + bmax(NEFES_H3BB_L1);
            save_trunk(1, A, es_a);
        }
        // even layers 2, 4, 6, 8 (A -> B); a macro so that the sigma-only kernel can run layer 8 behind the loop: with a `break` in
        // the middle of the rolled loop hipcc copied a whole accumulator set between register classes at the loop head
#define NEFES_FWD_EVEN_LAYER(L1_, SEG1_)                                                                                       \
        clear_bits();                                                                                                          \
        {                                                                                                                      \
            const int ew = wexp(SEG1_), tau = tau_of(M, ew);                                                                   \
            const float rb_ = rowb(SEG1_), bm_ = bmax((L1_) - 1);       /* table reads in front of the run, not behind it */    \
            float mx = 0.f;                                                                                                    \
            if constexpr (CB) {                                                                                                \
                const char* bps = bias_half + ((L1_) - 2) * W * 4;                          /* bias block of the source layer */ \
                nbias = ReluBiasSplitH<CAP, NTW, WT>::prime(bps);                                                              \
                mma_run_h3<NTW, W / 16, 0, true>(ring, ring_lane, ReluBiasSplitH<CAP, NTW, WT>{A, bits, pow2i(tau - es_a), mx, bps, pow2i(es_a), nbias}, \
                                                 ZeroInit{}, B);                                                               \
            } else                                                                                                             \
            mma_run_h3<NTW, W / 16, 0, true>(ring, ring_lane, ReluSplitH<CAP, NTW, WT>{A, bits, pow2i(tau - es_a), mx},        \
                                             bias_at(((L1_) - 1) * W, tau + ew), B);                                           \
            M = rb_ * (pair_max(mx) * pow2i(-es_a)) + bm_;                                                                     \
            es_b = tau + ew;                                                                                                   \
            save_trunk(L1_, B, es_b);                                                                                          \
        }                                                                                                                      \
        put_masks(bits, WT);                                                                      /* mask of layer l1-1 */
        constexpr int NPAIRS = MODE == NEFES_FIELD_SIGMA ? 3 : 4;
#pragma unroll 1
        for (int p = 0; p < NPAIRS; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            const int seg1 = l1 <= 5 ? l1 - 1 : l1;                                    // layout.h NEFES_H3F_*: L5 is two segments
            const int seg2 = l2 <= 5 ? l2 - 1 : (l2 <= 8 ? l2 : NEFES_H3F_FINAL);
            NEFES_FWD_EVEN_LAYER(l1, seg1)
            clear_bits();
            {
                const int ew = wexp(seg2);
                // skip layer: its xyz part accumulates into the same tiles, so the common exponent must suit the embedding too
                const int tau = tau_of(p == 1 ? fmaxf(M, mE) : M, ew);
                if (p == 3) sigma_head(B, es_b, tau_of(M, wexp(NEFES_H3F_SIG)));   // static_sigma reads the same relu(h8)
                const float rb_ = rowb(seg2), bm_ = bmax(l2 <= 8 ? l2 - 1 : NEFES_H3BB_FINAL);      // (table reads in front of the run)
                const float rbe_ = p == 1 ? rowb(NEFES_H3F_L5E) : 0.f;
                float mx = 0.f;
                if constexpr (CB) {
                    const char* bps = bias_half + (l1 - 1) * W * 4;                        // bias block of the source layer l1
                    nbias = ReluBiasSplitH<CAP, NTW, WT>::prime(bps);
                    const ReluBiasSplitH<CAP, NTW, WT> src{B, bits, pow2i(tau - es_b), mx, bps, pow2i(es_b), nbias};
                    mma_run_h3<NTW, W / 16, 0, true>(ring, ring_lane, src, ZeroInit{}, A);       // 3, 5, 7, final: zero start, bias at the consumers
                } else
                mma_run_h3<NTW, W / 16, 0, true>(ring, ring_lane, ReluSplitH<CAP, NTW, WT>{B, bits, pow2i(tau - es_b), mx},
                                                 bias_at(l2 <= 8 ? (l2 - 1) * W : B_FINAL, tau + ew), A);   // 3, 5, 7, final
                if (p == 1) mma_run_h3<NTW, ES / 8, 0, false>(ring, ring_lane, LdsSplitH{e_lds, pow2i(tau)}, ZeroInit{}, A);   // skip: + W5[:, :63] e
                M = rb_ * (pair_max(mx) * pow2i(-es_b)) + (p == 1 ? rbe_ * mE : 0.f) + bm_;
                es_a = tau + ew;
                save_trunk(l2, A, es_a);
            }
            put_masks(bits, WT);                                                                      // mask of layer l1
        }
        if constexpr (MODE == NEFES_FIELD_SIGMA) { NEFES_FWD_EVEN_LAYER(8, NEFES_H3F_L8) }
#undef NEFES_FWD_EVEN_LAYER
        if constexpr (MODE == NEFES_FIELD_SIGMA) sigma_head(B, es_b, tau_of(M, wexp(NEFES_H3F_SIG)));
        if constexpr (MODE != NEFES_FIELD_SIGMA) {
            // FULL: dir_encoding and transient_encoding.0 as ONE stacked 2*NTH-tile product (pack.cpp add_heads_x6): tiles
            // [0, NTH) = dir, [NTH, 2 NTH) = t0.  STATIC: dir_encoding alone (add_static_head_h3), NTH tiles.
            constexpr bool FULL = MODE == NEFES_FIELD_FULL;
            constexpr int NDT = FULL ? 2 * NTH : NTH;
            float v[3];
            {
                uint32_t mm, rr, ss;
                locate(mm, rr, ss);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[(size_t)rr * 3 + c];
            }
            float Dv[16];                                              // 14 embedding slots + 2 padding slots (two 16-k steps)
            {
                float d14[NEFES_D_STEPS];
                embed_slots<NEFES_N_FREQ_DIR>(d14, v, h);
#pragma unroll
                for (int s = 0; s < 16; ++s) Dv[s] = s < NEFES_D_STEPS ? d14[s] : 0.f;
            }
            if constexpr (TRAIN) {
                float* pd = act_tile + (size_t)(nefes_train_row(W, 0, NEFES_TB_DV) >> 5) * 4096 + emb_off;
#pragma unroll
                for (int s = 0; s < NEFES_D_STEPS; ++s) __builtin_nontemporal_store(Dv[s], &pd[2 * s * 16]);
            }
            f32x16 dt[NDT];
            uint32_t bits2[WH];
            auto clear2 = [&]() {
#pragma unroll
                for (int w = 0; w < WH; ++w) bits2[w] = 0u;
            };
            struct Bias2 {             // C operands of the stacked product: dir bias tiles, then t0 bias tiles
                BiasInitScaled a, b;
                __device__ __forceinline__ f32x16 operator()(int t) const { return t < NTH ? a(t) : b(t - NTH); }
            };
            int es_dt;
            {
                const int ew = wexp(NEFES_H3F_DT_H);                                   // = wexp(DT_D): one matrix
                const float mD = pair_max(array_max(Dv));                            // direction embedding (<= 1 for unit view dirs)
                const int tau = tau_of(fmaxf(M, mD), ew);                            // common exponent of both parts
                float mx = 0.f;
                es_dt = tau + ew;
                if constexpr (CB) {
                    const char* bpf = bias_half + B_FINAL * 4;                          // xyz_encoding_final's tiles come without its bias
                    nbias = ReluBiasSplitH<false, 1, 1>::prime(bpf);
                    const IdentBiasSplitH<NTW> srcf{A, pow2i(tau - es_a), mx, bpf, pow2i(es_a), nbias};
                    if constexpr (FULL) mma_run_h3<NDT, W / 16, 0, true>(ring, ring_lane, srcf, Bias2{bias_at(B_DIR, es_dt), bias_at(B_T0, es_dt)}, dt);
                    else mma_run_h3<NDT, W / 16, 0, true>(ring, ring_lane, srcf, bias_at(B_DIR, es_dt), dt);
                } else if constexpr (FULL)
                    mma_run_h3<NDT, W / 16, 0, true>(ring, ring_lane, IdentSplitH<NTW, 0>{A, pow2i(tau - es_a), mx},
                                                     Bias2{bias_at(B_DIR, es_dt), bias_at(B_T0, es_dt)}, dt);
                else
                    mma_run_h3<NDT, W / 16, 0, true>(ring, ring_lane, IdentSplitH<NTW, 0>{A, pow2i(tau - es_a), mx}, bias_at(B_DIR, es_dt), dt);
                mma_run_h3<NDT, 2, 0, false>(ring, ring_lane, ArraySplitH<16>{Dv, pow2i(tau)}, ZeroInit{}, dt);
                M = rowb(NEFES_H3F_DT_H) * (pair_max(mx) * pow2i(-es_a)) + rowb(NEFES_H3F_DT_D) * mD
                    + (FULL ? fmaxf(bmax(NEFES_H3BB_DIR), bmax(NEFES_H3BB_T0)) : bmax(NEFES_H3BB_DIR));   // both halves of the stacked output
                if constexpr (TRAIN) {
                    train_save_h3<NTH, 0>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_DIR), dt, pow2i(-es_dt));
                    if constexpr (FULL) train_save_h3<NTH, NTH>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_T0), dt, pow2i(-es_dt));
                }
            }
            if constexpr (FH) {
                // g = relu(dir_encoding output) in true units -> raw channels 3 + feature (feature 32 t + rho_h(r)); ones -> channel 3 + W/2
                float* col = raw_col();
                if (col) {
                    int S_g = a.S;
                    asm volatile("" : "+s"(S_g));
                    const float inv = pow2i(-es_dt);
                    float* pg = col + (size_t)(3 + 4 * h) * S_g;
#pragma unroll
                    for (int t = 0; t < NTH; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_nontemporal_store(fmaxf(dt[t][r] * inv, 0.f), &pg[(size_t)(32 * t + nefes_rho(0, r)) * S_g]);
                    if (h == 0) __builtin_nontemporal_store(1.f, &col[(size_t)(3 + W / 2) * S_g]);
                }
            }
            {
                f32x16 ar[NTR];
                clear2();
                const int ew = wexp(NEFES_H3F_RGB), tau = tau_of(M, ew), es = tau + ew;
                float mdummy = 0.f;
                mma_run_h3<NTR, W / 32, 0, true>(ring, ring_lane, ReluSplitH<true, NDT, WH>{dt, bits2, pow2i(tau - es_dt), mdummy},
                                                 bias_at(B_RGB, es), ar);
                put_masks(bits2, WH);                                 // dir_encoding
                float* col = raw_col();
                if (col) {
                    float* ph = col + (size_t)(4 * h) * a.S;
                    const float inv = pow2i(-es);
                    // C is a run-time value inside the head class: tiles whose 32 rows are all channels store unpredicated (a
                    // wave-uniform test per tile), the tile that holds row 3 + C carries a per-lane predicate, the padding tiles
                    // behind it store nothing (80 predicated stores per tile at C = 128 otherwise)
                    int S_t = a.S, c3 = FH ? 3 : 3 + a.C;         // opaque per tile: the 16 NTR row offsets cu * S and the channel tests
                    asm volatile("" : "+s"(S_t), "+s"(c3));      // are recomputed by the scalar ALU here instead of being hoisted out
                    const int full = c3 >> 5;                     // of the tile loop into (spilled) registers
#pragma unroll
                    for (int t = 0; t < NTR; ++t) {
                        if (t < full) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                __builtin_nontemporal_store(ar[t][r] * inv, &ph[(size_t)(32 * t + nefes_rho(0, r)) * S_t]);
                        } else if (32 * t < c3) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int cu = 32 * t + nefes_rho(0, r);
                                if (cu + 4 * h < c3) __builtin_nontemporal_store(ar[t][r] * inv, &ph[(size_t)cu * S_t]);
                            }
                        }
                    }
                }
            }
            if constexpr (FULL) {
            f32x16 acc3[NTH], acc2[NTH];
            int es3, es2, es_th;
            clear2();
            {
                const int ew = wexp(NEFES_H3F_T1), tau = tau_of(M, ew);
                float mx = 0.f;
                es3 = tau + ew;
                mma_run_h3<NTH, W / 32, 0, true>(ring, ring_lane, ReluSplitH<true, NDT, WH, NTH>{dt, bits2, pow2i(tau - es_dt), mx},
                                                 bias_at(B_T1, es3), acc3);
                M = rowb(NEFES_H3F_T1) * (pair_max(mx) * pow2i(-es_dt)) + bmax(NEFES_H3BB_T1);
                if constexpr (TRAIN) train_save_h3<NTH, 0>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_T1), acc3, pow2i(-es3));
            }
            put_masks(bits2, WH);                                     // transient_encoding.0
            clear2();
            {
                const int ew = wexp(NEFES_H3F_T2), tau = tau_of(M, ew);
                float mx = 0.f;
                es2 = tau + ew;
                mma_run_h3<NTH, W / 32, 0, true>(ring, ring_lane, ReluSplitH<true, NTH, WH>{acc3, bits2, pow2i(tau - es3), mx},
                                                 bias_at(B_T2, es2), acc2);
                M = rowb(NEFES_H3F_T2) * (pair_max(mx) * pow2i(-es3)) + bmax(NEFES_H3BB_T2);
                if constexpr (TRAIN) train_save_h3<NTH, 0>(act_tile, act_voff, nefes_train_row(W, 0, NEFES_TB_T2), acc2, pow2i(-es2));
            }
            put_masks(bits2, WH);                                     // transient_encoding.2
            f32x16 th[1];
            clear2();
            {
                const int ew = wexp(NEFES_H3F_TH), tau = tau_of(M, ew);
                float mdummy = 0.f;
                es_th = tau + ew;
                mma_run_h3<1, W / 32, 0, true>(ring, ring_lane, ReluSplitH<true, NTH, WH>{acc2, bits2, pow2i(tau - es2), mdummy},
                                               bias_at(B_TH, es_th), th);
            }
            put_masks(bits2, WH);
            float* col = raw_col();
            if (col) {
                float* o = col + (size_t)(3 + a.C + 1) * a.S;
                const float inv = pow2i(-es_th);
                if (h == 0) {
                    __builtin_nontemporal_store(sigmoid_ref(th[0][0] * inv), &o[0]);
                    __builtin_nontemporal_store(sigmoid_ref(th[0][1] * inv), &o[(size_t)a.S]);
                    __builtin_nontemporal_store(sigmoid_ref(th[0][2] * inv), &o[(size_t)2 * a.S]);
                    __builtin_nontemporal_store(softplus_ref(th[0][3] * inv), &o[(size_t)3 * a.S]);
                } else {
                    __builtin_nontemporal_store(softplus_ref(th[0][0] * inv), &o[(size_t)4 * a.S]);
                }
            }
            }
        }
    }
}

// magic multiplier for unsigned division by d, exact for dividends below 2^31: q = mulhi(n, magic) >> shift
static void magic_div(uint32_t d, uint32_t& magic, uint32_t& shift) {
    if (d == 1) { magic = 0; shift = 0; return; }               // the kernel takes magic == 0 as "quotient = dividend"
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                               // l = ceil(log2 d), 1..31
    const uint64_t p = 31 + l;                                 // n < 2^31: error bound 2^31 / 2^p * ... <= 1/d
    magic = (uint32_t)(((1ull << p) + d - 1) / d);             // ceil(2^p / d) < 2^32 because d > 2^(l-1)
    shift = (uint32_t)(p - 32);
}

template <int MODE, int ENC, int W = 256, int NTR = 1, bool TRAIN = false, bool FH = false>
static int launch_h3(const FieldFwdH3Args& a, hipStream_t st) {
    constexpr int ES_ = ENC != NEFES_XYZ_FREQ10 ? NEFES_X_STEPS : NEFES_E_STEPS;
    const size_t lds = (size_t)NEFES_H3_SLOTS * NEFES_SLAB_BYTES + ((a.bias_floats + 63) / 64) * 256 + (size_t)4 * ES_ * 64 * 4
                       + (ENC == NEFES_XYZ_HASHGRID_FUSED ? 512 : 0);
    static_assert(sizeof(HgGeom) <= 512, "the level geometry's LDS slot");
    auto k = field_fwd_h3_kernel<MODE, ENC, W, NTR, TRAIN, FH>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // Wd = 128: the kernel fits 256 registers and 70 KiB of LDS, so two workgroups share a CU -- two waves per SIMD, one's vector
    // work (6 VALU per MFMA at this width) under the other's MFMAs
    const int per_cu = W == 128 ? 2 : 1;
    int grid = a.n_tiles < cus * per_cu ? a.n_tiles : cus * per_cu;
    if (const char* cap = getenv("NEFES_DEBUG_MAX_GRID")) {          // experiments only: fewer workgroups than CUs
        const int c = atoi(cap);
        if (c > 0 && c < grid) grid = c;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

// Kernel instances spread over nine objects built from this one source (Makefile: -DNEFES_TU_PART=0..8; even parts from 2 on are
// the Wd = 128 objects): part 0 = entry points + Wd = 256 / head class 0, part 1 = hash-grid instances, part 2 = Wd = 128 / class 1,
// parts 3 / 4 = their TRAIN instances, parts 5 / 6 = Wd = 256 / class 1 and Wd = 128 / class 0, parts 7 / 8 = their TRAIN instances.
#ifndef NEFES_TU_PART
#define NEFES_TU_PART 0
#endif
enum { H3_EXT_SIGMA = 0, H3_EXT_FULL, H3_SIGMA, H3_FULL, H3_TRAIN_STATIC, H3_TRAIN_FULL, H3_STATIC, H3_HG_SIGMA, H3_HG_FULL, H3_FH_FULL };
int nefes_fwd_h3_launch_part1(int which, const FieldFwdH3Args& a, hipStream_t st);
int nefes_fwd_h3_launch_part2(int which, const FieldFwdH3Args& a, hipStream_t st);
int nefes_fwd_h3_launch_part3(int which, const FieldFwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 256, class 0
int nefes_fwd_h3_launch_part4(int which, const FieldFwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 128, class 1
int nefes_fwd_h3_launch_part5(int which, const FieldFwdH3Args& a, hipStream_t st);   // Wd = 256, class 1 (the reference's FEATURE_DIM at netwidth 256)
int nefes_fwd_h3_launch_part6(int which, const FieldFwdH3Args& a, hipStream_t st);   // Wd = 128, class 0
int nefes_fwd_h3_launch_part7(int which, const FieldFwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 256, class 1
int nefes_fwd_h3_launch_part8(int which, const FieldFwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 128, class 0


#if NEFES_TU_PART == 1
int nefes_fwd_h3_launch_part1(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_EXT_SIGMA: return launch_h3<NEFES_FIELD_SIGMA, NEFES_XYZ_EXTERNAL32>(a, st);
        case H3_EXT_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_EXTERNAL32>(a, st);
        case H3_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 256, 1>(a, st);    // static head alone, inference (round 5)
        case H3_HG_SIGMA: return launch_h3<NEFES_FIELD_SIGMA, NEFES_XYZ_HASHGRID_FUSED>(a, st);     // hash grid gathered in the prologue (round 5)
        case H3_HG_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_HASHGRID_FUSED>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 2
// The Wd = 128 instances hold 2 x 4 accumulator tiles = 128 registers: with ~120 more for everything else the whole kernel fits
// the 256 architectural VGPRs, and this object is built with -mllvm -amdgpu-mfma-vgpr-form (Makefile) so that the MFMAs
// accumulate there.  Left to its heuristics hipcc parks the tiles in AGPRs and pays a v_accvgpr_read / _write for every value
// the vector ALU touches (1.7 of the kernel's 6.3 VALU per MFMA: it is VALU-bound at this width): forward 0.85 -> 0.78 ms,
// backward 0.77 -> 0.73 ms on the 80x60 refinement frame.
int nefes_fwd_h3_launch_part2(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_SIGMA: return launch_h3<NEFES_FIELD_SIGMA, NEFES_XYZ_FREQ10, 128, 5>(a, st);       // (the sigma-only pass has no rgb head: one instance per width)
        case H3_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 5>(a, st);
        case H3_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 128, 5>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 3
int nefes_fwd_h3_launch_part3(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_TRAIN_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 256, 1, true>(a, st);
        case H3_TRAIN_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 256, 1, true>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 4      // (built like part 2)
int nefes_fwd_h3_launch_part4(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_TRAIN_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 128, 5, true>(a, st);
        case H3_TRAIN_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 5, true>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 5
int nefes_fwd_h3_launch_part5(int which, const FieldFwdH3Args& a, hipStream_t st) {
    if (which == H3_FULL) return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 256, 5>(a, st);
    if (which == H3_STATIC) return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 256, 5>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 6      // (built like part 2)
int nefes_fwd_h3_launch_part6(int which, const FieldFwdH3Args& a, hipStream_t st) {
    if (which == H3_FULL) return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 1>(a, st);
    if (which == H3_STATIC) return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 128, 1>(a, st);
    if (which == H3_FH_FULL) return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 1, false, true>(a, st);      // factored head (round 5)
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 7
int nefes_fwd_h3_launch_part7(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_TRAIN_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 256, 5, true>(a, st);
        case H3_TRAIN_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 256, 5, true>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 8      // (built like part 2)
int nefes_fwd_h3_launch_part8(int which, const FieldFwdH3Args& a, hipStream_t st) {
    switch (which) {
        case H3_TRAIN_STATIC: return launch_h3<NEFES_FIELD_STATIC, NEFES_XYZ_FREQ10, 128, 1, true>(a, st);
        case H3_TRAIN_FULL: return launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 1, true>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#else   // part 0

// Train-mode forward on the fp16 pipe: as nefes_field_fwd_train (field_fwd.hip), same `acts` rows, same masks, same raw_t.
extern "C" int nefes_field_fwd_train_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                        const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                        float* raw_t, float* acts, uint32_t* masks, void* stream) {
    if (!desc || !packed || !raw_t || !acts || !viewdirs || N <= 0 || S <= 0) return NEFES_E_BADARG;
    if (!pts && !(rays_o && rays_d && z)) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_STATIC && mode != NEFES_FIELD_FULL) return NEFES_E_UNSUPPORTED;
    if (mode == NEFES_FIELD_FULL && !desc->has_transient) return NEFES_E_BADARG;
    const int cls = nefes_head_class(desc->feat_dim);
    if ((desc->width != 256 && desc->width != 128) || cls < 0 || desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[mode == NEFES_FIELD_STATIC ? NEFES_STREAM_FWD_STATIC_H3 : NEFES_STREAM_FWD_FULL_H3];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldFwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.bias = (const float*)((const char*)packed + si.bias_off);
    a.n_slabs = si.n_slabs; a.bias_floats = si.bias_floats; a.scale_off = si.scale_off;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.xyz_enc = nullptr; a.viewdirs = viewdirs; a.raw_t = raw_t; a.masks = masks;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + (mode == NEFES_FIELD_STATIC ? 1 : 6);
    a.M = (long long)N * S;
    if (a.M >= (1ll << 31) - 256) return NEFES_E_UNSUPPORTED;      // the kernel indexes samples with 32 bits
    a.n_tiles = (int)((a.M + 127) / 128);
    a.acts = acts;
    a.rows = nefes_train_row(desc->width, desc->feat_dim, NEFES_TB_END);
    a.z_row = 0; a.gout = 0; a.hg_table = nullptr;
    magic_div((uint32_t)S, a.s_magic, a.s_shift);
    const int which = mode == NEFES_FIELD_STATIC ? H3_TRAIN_STATIC : H3_TRAIN_FULL;
    hipStream_t st = (hipStream_t)stream;
    if (desc->width == 256) return cls == 0 ? nefes_fwd_h3_launch_part3(which, a, st) : nefes_fwd_h3_launch_part7(which, a, st);
    return cls == 1 ? nefes_fwd_h3_launch_part4(which, a, st) : nefes_fwd_h3_launch_part8(which, a, st);
}

static int field_fwd_h3_impl(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                             const float* rays_d, const float* z, int z_row, const float* pts, const float* xyz_enc,
                             const float* viewdirs, float* raw_t, uint32_t* masks, void* stream,
                             const NefesHashGridDesc* grid = nullptr, const float* table = nullptr, bool fh = false) {
    if (!desc || !packed || !raw_t || N <= 0 || S <= 0) return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    const bool fused_grid = ext && table != nullptr;            // the kernel gathers the 32 features itself (hashgrid.h)
    if (fused_grid ? !(rays_o && rays_d && z && grid) : (ext ? !xyz_enc : (!pts && !(rays_o && rays_d && z)))) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_SIGMA && mode != NEFES_FIELD_FULL && mode != NEFES_FIELD_STATIC) return NEFES_E_UNSUPPORTED;
    if (mode == NEFES_FIELD_FULL && (!viewdirs || !desc->has_transient)) return NEFES_E_BADARG;
    if (mode == NEFES_FIELD_STATIC && (!viewdirs || ext)) return ext ? NEFES_E_UNSUPPORTED : NEFES_E_BADARG;   // (frequency embedding only)
    // compiled set: widths 128 / 256 x head classes 0 / 1 (layout.h) with the frequency embedding; width 256 / class 0 with an
    // external 32-feature embedding
    const int cls = nefes_head_class(desc->feat_dim);
    const bool big = desc->width == 256, small = desc->width == 128 && !ext;
    if (!(big || small) || cls < 0 || (ext && cls != 0) || (desc->xyz_encoding != NEFES_XYZ_FREQ10 && !ext)) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[mode == NEFES_FIELD_SIGMA ? NEFES_STREAM_FWD_SIGMA_H3
                                            : (mode == NEFES_FIELD_STATIC ? NEFES_STREAM_FWD_STATIC_H3 : NEFES_STREAM_FWD_FULL_H3)];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldFwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.bias = (const float*)((const char*)packed + si.bias_off);
    a.n_slabs = si.n_slabs; a.bias_floats = si.bias_floats; a.scale_off = si.scale_off;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.xyz_enc = xyz_enc; a.viewdirs = viewdirs; a.raw_t = raw_t; a.masks = masks;
    a.acts = nullptr; a.rows = 0; a.z_row = z_row;
    a.hg_table = (const float2*)table;
    if (fused_grid) {
        rc = hg_geometry(grid, &a.hg, nullptr);
        if (rc) return rc;
        if (a.hg.n_levels != 16) return NEFES_E_UNSUPPORTED;       // sixteen levels x two features = the network's 32 inputs
    }
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = mode == NEFES_FIELD_SIGMA ? 1 : 3 + a.C + (mode == NEFES_FIELD_STATIC ? 1 : 6);
    a.gout = 0;
    if (fh) {
        // factored head: a network packed WITHOUT its feature rows (feat_dim 0: the 3-row colour head) whose raw output carries
        // g = relu(dir_encoding) and a channel of ones where the feature channels would be: C' = W/2 + 1 "feature" channels
        if (desc->feat_dim != 0 || desc->width != 128 || ext || mode != NEFES_FIELD_FULL) return NEFES_E_UNSUPPORTED;
        a.gout = desc->width / 2;
        a.C = a.gout + 1;
        a.R = 3 + a.C + 6;
    }
    a.M = (long long)N * S;
    if (a.M >= (1ll << 31) - 256) return NEFES_E_UNSUPPORTED;      // the kernel indexes samples with 32 bits
    a.n_tiles = (int)((a.M + 127) / 128);
    magic_div((uint32_t)S, a.s_magic, a.s_shift);
    hipStream_t st = (hipStream_t)stream;
    if (fh) return nefes_fwd_h3_launch_part6(H3_FH_FULL, a, st);
    if (mode == NEFES_FIELD_STATIC) {
        // The static head alone at inference (round 5): what a frozen coarse network runs when test_time is False (rendering.py:116-125)
        // and a fine network with NeRFW off (nerfh_nff.py:217-231 with output_transient False) -- the TRAIN instances' kernel without
        // the activation stores.
        if (big) return cls == 0 ? nefes_fwd_h3_launch_part1(H3_STATIC, a, st) : nefes_fwd_h3_launch_part5(H3_STATIC, a, st);
        return cls == 1 ? nefes_fwd_h3_launch_part2(H3_STATIC, a, st) : nefes_fwd_h3_launch_part6(H3_STATIC, a, st);
    }
    if (fused_grid) return nefes_fwd_h3_launch_part1(mode == NEFES_FIELD_SIGMA ? H3_HG_SIGMA : H3_HG_FULL, a, st);
    if (ext) return nefes_fwd_h3_launch_part1(mode == NEFES_FIELD_SIGMA ? H3_EXT_SIGMA : H3_EXT_FULL, a, st);
    if (small) {
        if (mode == NEFES_FIELD_SIGMA) return nefes_fwd_h3_launch_part2(H3_SIGMA, a, st);
        return cls == 1 ? nefes_fwd_h3_launch_part2(H3_FULL, a, st) : nefes_fwd_h3_launch_part6(H3_FULL, a, st);
    }
    if (mode == NEFES_FIELD_SIGMA) return launch_h3<NEFES_FIELD_SIGMA, NEFES_XYZ_FREQ10>(a, st);
    return cls == 0 ? launch_h3<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10>(a, st) : nefes_fwd_h3_launch_part5(H3_FULL, a, st);
}

extern "C" int nefes_field_fwd_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* xyz_enc,
                                  const float* viewdirs, float* raw_t, uint32_t* masks, void* stream) {
    return field_fwd_h3_impl(desc, packed, mode, N, S, rays_o, rays_d, z, 0, pts, xyz_enc, viewdirs, raw_t, masks, stream);
}

// A NEFES_XYZ_EXTERNAL32 network whose 32 features are a multiresolution hash grid of pts = o + d z (BASELINE configs[3]:
// script/models/nerfh_tcnn.py:60-75,151-182): the kernel gathers the table itself instead of reading an [M, 32] encoding that
// nefes_hashgrid_fwd wrote (10 GB per fine pass at 854x480).  z: [N, S], or one row [S] shared by every ray (z_is_row).
extern "C" int nefes_field_fwd_h3_hashgrid(const NefesNetDesc* desc, const void* packed, const NefesHashGridDesc* grid,
                                           const float* table, int mode, int N, int S, const float* rays_o, const float* rays_d,
                                           const float* z, int z_is_row, const float* viewdirs, float* raw_t, uint32_t* masks,
                                           void* stream) {
    if (!desc || desc->xyz_encoding != NEFES_XYZ_EXTERNAL32 || !grid || !table) return NEFES_E_BADARG;
    return field_fwd_h3_impl(desc, packed, mode, N, S, rays_o, rays_d, z, z_is_row ? 1 : 0, nullptr, nullptr, viewdirs, raw_t, masks, stream,
                             grid, table);
}

// Factored head (see the kernel's FH parameter): `desc` / `packed` describe the network WITHOUT its feature rows (feat_dim 0);
// raw_t [N][3 + (W/2 + 1) + 6][S] = rgb (3) | relu(dir_encoding) (W/2) | ones | sigma | transient rgb (3), sigma, beta.
extern "C" int nefes_field_fwd_h3_fh(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                     const float* rays_d, const float* z, const float* viewdirs, float* raw_t, uint32_t* masks,
                                     void* stream) {
    if (!rays_o || !rays_d || !z) return NEFES_E_BADARG;
    return field_fwd_h3_impl(desc, packed, mode, N, S, rays_o, rays_d, z, 0, nullptr, nullptr, viewdirs, raw_t, masks, stream, nullptr, nullptr, true);
}

// The same pass with ONE row of S depths shared by every ray (`z_row` [S]): the coarse pass at test time with scalar near / far
// (rendering.py:96-100) -- the [N, S] depth tensor is then never materialised (79 MB at the headline shape, and a launch).
extern "C" int nefes_field_fwd_h3_zrow(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                       const float* rays_d, const float* z_row, const float* viewdirs, float* raw_t, uint32_t* masks,
                                       void* stream) {
    if (!rays_o || !rays_d || !z_row) return NEFES_E_BADARG;
    if (desc && desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
    return field_fwd_h3_impl(desc, packed, mode, N, S, rays_o, rays_d, z_row, 1, nullptr, nullptr, viewdirs, raw_t, masks, stream);
}
#endif   // NEFES_TU_PART
