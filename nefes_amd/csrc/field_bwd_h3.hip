// Fused field backward-to-inputs with the hidden products as fp16 two-part split products (field_h3.h): what autograd does in
// the reference for d raw -> d pts, d viewdirs through NeRFH_NFF.forward and Embedder.embed (script/models/nerfh_nff.py:525-576,
// :234-270) with frozen weights.  Same chain of W^T products, same inputs / outputs / ReLU-mask words as field_bwd_kernel
// (field_bwd.hip); the transposed products of transient_encoding.{4,2,0}, dir_encoding, xyz_encoding_final and layers 8..1 run on
// v_mfma_f32_32x32x16_f16 as hh + hl + lh of power-of-two scaled (hi, lo) fp16 pairs; the three narrow head products (3+C, 5 and
// 1 k-values) stay on the fp32 MFMA.  Every gradient vector carries a per-lane scale exponent (field_h3.h).
#if defined(NEFES_TU_PART) && NEFES_TU_PART >= 2 && NEFES_TU_PART % 2 == 0
#define NEFES_SLAB_KIB 16      // even parts from 2 on: the Wd = 128 instances (layout.h NEFES_H3_BWD_SLAB_KIB_128)
#define NEFES_ASM_READS_ACC    // built with -amdgpu-mfma-vgpr-form: the functors' asm statements read the MFMAs' own VGPRs (field_common.h)
#define NEFES_ACC_VGPR_FORM
#define NEFES_FENCE_TILES
#else
#define NEFES_SLAB_KIB 32
// The Wd = 256 inference objects (parts 0, 1) run their 8-/10-tile products on field_h3.h's gap-by-gap schedule (asm MFMAs on AGPR
// tiles, source tiles read out of their AGPRs inside the gaps), like the forward.  That needs the accumulator file to hold EXACTLY
// the 2 x NTW ping-pong tiles: with a seventeenth tile alive hipcc relocates whole tiles with v_accvgpr_mov in the middle of the asm
// runs -- behind an MFMA that has not written them yet.  So the three embedding-gradient tiles are consumed
// where they are produced (below), layer 5's two extra output tiles live in VGPRs for the length of that run, and
// tests/test_pack_stream.py disassembles the library and fails if a v_accvgpr_mov shows up in these kernels.
#if !defined(NEFES_TU_PART) || NEFES_TU_PART == 0 || NEFES_TU_PART == 1 || NEFES_TU_PART == 5
#define H3B_WIDE
#define H3_ACC_READ_ASM        // source tiles are read out of their AGPRs inside the MFMA gaps (field_h3.h acc_read)
#define NEFES_ASM_READS_ACC    // ... by asm statements: compiler-placed runs end with field_common.h mfma_results_fence_tiles
#define NEFES_FENCE_TILES
#define H3_WIDE_ENTRY_FENCE    // wide runs follow compiler-scheduled segments here (field_h3.h mma_run_h3_wide)
#else
#define NEFES_H3_WIDE_MIN 99   // TRAIN instances: block-per-pair form for every segment
#endif
#endif
#define NEFES_B_BATCH 2
#define NEFES_B_BATCH_NT8 4
#include "field_common.h"
#include "field_x6.h"
#include "field_h3.h"
#include "hashgrid.h"
#include "../../include/nefes_hip.h"
#define NEFES_XYZ_HASHGRID_FUSED 2   /* kernel-internal ENC value (field_fwd_h3.hip): the 32-feature hash-grid encoding evaluated by the kernel */
#ifndef H3B_WIDE_LAYERS
#define H3B_WIDE_LAYERS 0x1ff   /* bit L: layer L's transposed product on the gap-by-gap schedule; bit 0: xyz_encoding_final's (debugging) */
#endif
#ifndef NEFES_H3B_WG128
// Workgroups per CU of the Wd = 128 instances.  Rounds 2-3: 2 needed ~100 registers spilled to scratch and measured no gain.  Round 4:
// with C a run-time value the per-channel row offsets and channel tests of the tile's loads are recomputed per tile by the scalar ALU
// instead of being hoisted into ~150 (spilled) registers, the kernel needs 232-242 VGPRs, and two workgroups per CU -- one wave's
// vector work under the other's MFMAs, as in the forward -- measure 0.618 vs 0.684 ms on the 80x60 frame (A/B on one box against the
// round-3 library; the same source at one workgroup per CU: 0.72).
#define NEFES_H3B_WG128 2
#endif
#define NEFES_H3B_WG_PER_CU(W) ((W) == 128 ? NEFES_H3B_WG128 : 1)
#define NEFES_H3B_SLOTS 2   // 64 KiB ring (StagedRing: two slots) + the tile's ReLU masks and the scale table

struct FieldBwdH3Args {
    const char* stream;
    const int* tab;         // scale table of the stream (layout.h: per segment weight exponent + row bound), in the blob
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
    const float2* hg_table; // NEFES_XYZ_HASHGRID_FUSED: the hash-grid table and its level geometry (hashgrid.h)
    HgGeom hg;
    int gout;               // FH instances (field_fwd_h3.hip): W / 2 channels of d loss / d relu(dir_encoding) in the feature channels' place
    const float* g_gmap;    // FH, optional: [N][gout + 1] = d loss / d (composited g) per RAY; d loss / d g of a sample is then formed here as
                            // w_s g_gmap[ray][f], with the static weight w_s read from the first feature channel's row of g_raw_t
                            // (nefes_composite_bwd with NEFES_COMP_FEAT_WEIGHTS_ONLY) instead of gout rows of products
};

// TRAIN instances: the (masked) gradient vector a product consumes is d loss / d pre-activation of a hidden layer, which the
// weight-gradient kernels (train.hip) need: it is stored on the way in, in true units (inv = 2^-exponent of the source tiles).
// Element e = 8q + 2p (+1) of this lane's sample <-> feature 32 (e / 16) + rho_h(e % 16): row p[...] of the block (p carries the
// block's first row, the lane half's +4 rows and the sample column) -- as Storing / StoringSplit of the fp32 / bf16x6 kernels.
template <class Inner>
struct StoringSplitH {
    Inner in;
    float* p;
    float inv;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int pp) const {
        in.stage_a(s, q, pp);
        const int e = 8 * q + 2 * pp;
#ifndef NEFES_TRAIN_NO_STORES      /* (ablation builds: what the stores cost, tools/ab_side.sh) */
        __builtin_nontemporal_store(s.x0 * inv, &p[(e >> 4) * 4096 + nefes_rho(0, e & 15) * 16]);      // layout.h nefes_train_off
        __builtin_nontemporal_store(s.x1 * inv, &p[((e + 1) >> 4) * 4096 + nefes_rho(0, (e + 1) & 15) * 16]);
#endif
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const { in.stage_b(s); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int pp, const PairRegs& s) const { in.template stage_c<NOP>(o, pp, s); }
    __device__ __forceinline__ void stage_c1(Split2& o, int pp, const PairRegs& s) const { in.stage_c1(o, pp, s); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int pp, const PairRegs& s) const { in.template stage_c2<NOP>(o, pp, s); }
};
template <bool ON, class Inner>
__device__ __forceinline__ auto wrap_store_h3(const Inner& in, float* p, float inv) {
    if constexpr (ON) return StoringSplitH<Inner>{in, p, inv};
    else return in;
}

// Streaming inputs of a tile (ReLU masks, raw outputs, upstream gradient: read once, 27 GB per headline launch) are loaded
// non-temporally: as ordinary loads they pushed the 2.6 MB weight stream, which every workgroup re-reads per tile, out of the L2s --
// the HBM counters showed 63 GB fetched per launch (FETCH_SIZE x 2 is exact for these load widths: tools/probe/fetch_probe.hip).
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
    return __builtin_nontemporal_load(p);
}

// HAS_T = false: the static head only (NEFES_FIELD_STATIC forward: raw channels rgb+feature, sigma); the stream then is
// NEFES_STREAM_BWD_STATIC_H3 and the transient segments are absent.  TRAIN: see StoringSplitH.
// KR16 = k-steps of 16 upstream channels of static_rgb^T = the head class of layout.h (2: 3 + C <= 32; 9: 3 + C <= 144); C itself is
// a run-time argument (a.C).  ENC = NEFES_XYZ_*
// FH: the factored head of field_fwd_h3.hip -- the upstream gradient carries d loss / d g (g = relu(dir_encoding)) in raw channels
// 3 .. 3 + W/2, which joins the 3-row colour head's transposed product in front of dir_encoding^T.
template <int W, int KR16, int ENC, bool HAS_T = true, bool TRAIN = false, bool FH = false>
__global__ __launch_bounds__(256, NEFES_H3B_WG_PER_CU(W)) void field_bwd_h3_kernel(FieldBwdH3Args a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128);
    constexpr int WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;   // mask words per trunk / half-width layer
    constexpr int MW_TRUNK = 8 * WT;
    constexpr int NTR = KR16 <= 2 ? 1 : 5;                          // tiles of the rgb+feature head block in `dacts` (TRAIN)
    // Layers on the gap-by-gap schedule (asm MFMAs, field_h3.h mma_run_h3_wide).  Not in the instance with the hash grid in its
    // epilogue: there hipcc splits the live range of one accumulator tile INSIDE an asm-scheduled run whenever anything about the
    // kernel's register use changes (round 5: the skip's share parked in LDS; round 6: the ReLU-mask words read in place) -- a
    // v_accvgpr_mov one wait state behind the asm MFMA that writes the tile, where hipcc would put twelve behind an MFMA of its own
    // (tools/hazard_lint.py rule B1; tests/test_pack_stream.py finds the move itself; round 3 saw wrong gradients from such a move).  With compiler-placed MFMAs a moved tile is
    // the compiler's to pad.  Costs configs[3] ~4 ms of 590 per frame.
    constexpr int WIDE_LAYERS = ENC == NEFES_XYZ_HASHGRID_FUSED ? 0 : H3B_WIDE_LAYERS;
    static_assert(KR16 == 2 || KR16 == 9, "head classes of layout.h (nefes_head_kr16 / nefes_head_ntr)");
    const int C3 = 3 + a.C;                                         // static rgb/feature head^T: 3+C upstream channels as fp16 k-steps
    static_assert(MW % 4 == 0, "mask words are staged as 16-byte groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // per-lane LDS column behind the mask words: 8 stash words (x, v, d sigma); the fused hash-grid instance parks sixteen more there
    // (the skip's share of d encoding waits for layer 1's: in registers it pushed the allocator into moving accumulator tiles)
    constexpr int SX = ENC == NEFES_XYZ_HASHGRID_FUSED ? 24 : 8;
    int* tab_i = (int*)(smem + NEFES_H3B_SLOTS * NEFES_SLAB_BYTES + (size_t)4 * (MW + SX) * 256);
    const float* tab_f = (const float*)tab_i;
    if (threadIdx.x < 2 * (HAS_T ? NEFES_H3B_N : NEFES_H3B_N_STATIC)) tab_i[threadIdx.x] = a.tab[threadIdx.x];
    HgGeom* hg_lds = (HgGeom*)(tab_i + 64);                     // NEFES_XYZ_HASHGRID_FUSED: level geometry behind the scale table (see field_fwd_h3.hip)
    if constexpr (ENC == NEFES_XYZ_HASHGRID_FUSED) {
        if (threadIdx.x == 0) *hg_lds = a.hg;
    }
    auto wexp = [&](int seg) { return tab_i[nefes_h3_tab_exp(nefes_h3b_seg(HAS_T, seg))]; };     // (segment ordinals of the full stream)
    auto rowb = [&](int seg) { return tab_f[nefes_h3_tab_bound(nefes_h3b_seg(HAS_T, seg))]; };
    StagedRing ring;
    ring.init(a.stream, a.n_slabs, smem, wave, lane);
    const char* ring_lane = smem + lane * 16;
    ring.prime(ring_lane);                                       // (its barrier also publishes the scale table)
    // this wave's mask words in LDS: [MW/4][64 lanes][4 words]
    uint32_t* mlds = (uint32_t*)(smem + NEFES_H3B_SLOTS * NEFES_SLAB_BYTES) + wave * ((MW + SX) * 64) + lane * 4;
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

        // ================= the tile's global loads (the weight ring moves through registers here: the compiler counts all loads) =====
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
        float y_th[3] = {0.f, 0.f, 0.f}, g_th[3] = {0.f, 0.f, 0.f}, y_sg, g_sg, dr[8 * KR16];
        if constexpr (HAS_T) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {            // compact slot (s,h) <-> transient-head row 2s+h (5 rows)
                const int row = 2 * s + h;
                const int ch = cT + (row < 5 ? row : 4);
                y_th[s] = ld_stream(&a.raw_t[chan0 + (size_t)ch * a.S]);
                g_th[s] = ld_stream(&a.g_raw_t[chan0 + (size_t)ch * a.S]);
            }
        }
        y_sg = ld_stream(&a.raw_t[chan0 + (size_t)C3 * a.S]);
        g_sg = ld_stream(&a.g_raw_t[chan0 + (size_t)C3 * a.S]);
        // element e of this lane <-> static rgb/feature channel 32 (e / 16) + rho_h(e % 16): the natural slot order of the fp16 operands.
        // Addressing: the channel's row offset ch0 * S is wave-uniform (scalar ALU), the lane half's + 4 rows is one per-lane pointer.
        // S passes through an opaque asm once per tile: with C a run-time value hipcc otherwise hoists the 8 * KR16 per-lane 64-bit
        // row offsets out of the tile loop (144 registers at KR16 = 9, spilled to scratch -- seen in the disassembly).
        // (the same for 3 + C, whose 2 x 8 KR16 wave-uniform comparisons would be hoisted into spilled scalar registers)
        {
            int S_t = a.S, c3 = FH ? 3 : C3;                  // (FH: only the three colour channels feed the head's transposed product)
            asm volatile("" : "+s"(S_t), "+s"(c3));
            const float* g0 = a.g_raw_t + chan0;
            const float* g4 = g0 + (size_t)(4 * h) * S_t;
            // Per GROUP of sixteen elements (32 channels), not per element: a test per element put every load into its own basic block
            // with a select right behind it, i.e. 8 KR16 memory round trips one after the other at the top of every tile (72 at
            // KR16 = 9: seen in the disassembly as load / s_waitcnt vmcnt(0) / v_cndmask triplets).  A group whose 32 channels all
            // exist -- every group but the network's last -- is sixteen plain loads in flight together; the last, partial group reads
            // rows past the last channel as the last row (valid memory) and zeroes them once all its loads are out.
            const int h4 = 4 * h;
#pragma unroll
            for (int grp = 0; grp < (8 * KR16 + 15) / 16; ++grp) {
                const int e0 = 16 * grp, e1 = e0 + 16 < 8 * KR16 ? e0 + 16 : 8 * KR16;
                if (32 * grp + 32 <= c3) {
#pragma unroll
                    for (int e = e0; e < e1; ++e) dr[e] = ld_stream(g4 + (size_t)(32 * grp + nefes_rho(0, e & 15)) * S_t);
                } else if (32 * grp < c3) {
#pragma unroll
                    for (int e = e0; e < e1; ++e) {
                        const int row = 32 * grp + nefes_rho(0, e & 15) + h4;
                        dr[e] = ld_stream(g0 + (size_t)(row < c3 ? row : c3 - 1) * S_t);
                    }
#pragma unroll
                    for (int e = e0; e < e1; ++e) dr[e] = (32 * grp + nefes_rho(0, e & 15) + h4 < c3) ? dr[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = e0; e < e1; ++e) dr[e] = 0.f;
                }
            }
        }
        uint4 mq[MW / 4];
        {
            const uint32_t* mk32 = a.masks + ((size_t)(m >> 5) * MW) * 64 + lane;
#pragma unroll
            for (int q = 0; q < MW / 4; ++q) {
                mq[q].x = ld_stream(&mk32[(4 * q + 0) * 64]); mq[q].y = ld_stream(&mk32[(4 * q + 1) * 64]);
                mq[q].z = ld_stream(&mk32[(4 * q + 2) * 64]); mq[q].w = ld_stream(&mk32[(4 * q + 3) * 64]);
            }
        }
        // ======================================================================================================
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) *(uint4*)(mlds + q * 256) = mq[q];     // own lane's words only: no barrier needed
        if (!valid) {
#pragma unroll
            for (int s = 0; s < 3; ++s) g_th[s] = 0.f;
            g_sg = 0.f;
#pragma unroll
            for (int e = 0; e < 8 * KR16; ++e) dr[e] = 0.f;
        }
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
        const float d_sigma = h == 0 ? g_sg * (1.f - expf(-y_sg)) : 0.f;
        STASH(6) = d_sigma;
        if constexpr (TRAIN) {
            // The head blocks of the gradient buffer -- d raw for the rgb+feature channels, the sigma and transient-head
            // pre-activation gradients, padded with zero rows to whole 32-row tiles -- are what the weight-gradient kernels read as
            // G of static_rgb / static_sigma / the transient heads.  They are in registers here (compact slots: row 2s + h), so
            // they are stored here: the separate head-gradient pass (train.hip train_head_grad_kernel) re-read d raw and raw_t.
            float* dt_ = a.dacts + (size_t)tile * a.rows * 128 + (size_t)(((wave * 32 + j) >> 4) * 512 + h * 16 + (j & 15));
            float* prgb = a.dacts + (size_t)tile * a.rows * 128 + (size_t)(nefes_train_row(W, a.C, NEFES_TB_RGB) >> 5) * 4096 + nefes_train_lane_off(wave, j, h);
#pragma unroll
            for (int e = 0; e < 16 * NTR; ++e)      // row 32 (e / 16) + rho_h(e % 16)
                __builtin_nontemporal_store(e < 8 * KR16 ? dr[e < 8 * KR16 ? e : 0] : 0.f, &prgb[(e >> 4) * 4096 + nefes_rho(0, e & 15) * 16]);
            float* psig = dt_ + (size_t)(nefes_train_row(W, a.C, NEFES_TB_SIG) >> 5) * 4096;
#pragma unroll
            for (int s = 0; s < 16; ++s) __builtin_nontemporal_store(s == 0 ? d_sigma : 0.f, &psig[2 * s * 16]);
            if constexpr (HAS_T) {
                float* pth = dt_ + (size_t)(nefes_train_row(W, a.C, NEFES_TB_TH) >> 5) * 4096;
#pragma unroll
                for (int s = 0; s < 16; ++s) __builtin_nontemporal_store(s < 3 ? dth[s < 3 ? s : 0] : 0.f, &pth[2 * s * 16]);
            }
        }

        auto load_bits = [&](uint32_t* b, int word0, int n) {
            for (int w = 0; w < n; ++w) b[w] = MASKW(word0 + w);
        };
        uint32_t bh[WH], bt[WT];
        // Scale bookkeeping as in field_fwd_h3.hip: es_x = exponent an accumulator set carries, M = upper bound of a gradient
        // vector's largest magnitude in true units (packer's row bounds x the exactly measured maximum of the previous operand),
        // tau = exponent the operand is brought to; output exponent = tau + weight exponent.
        auto tau_of = [&](float M, int ew) {
            const int t = pick_exp(M);
            return t < 100 - ew ? t : 100 - ew;
        };
        // TRAIN: this lane's column of a hidden block in the gradient buffer (first row + 4 rows for lane half 1)
        auto gptr = [&](int block) -> float* {
            if constexpr (!TRAIN) return nullptr;
            else return a.dacts + (size_t)tile * a.rows * 128 + (size_t)(nefes_train_row(W, 0, block) >> 5) * 4096 + nefes_train_lane_off(wave, j, h);
        };
        f32x16 G2[NTH], T3[NTH], T4[NTH];
        // ---- static_rgb^T: 3+C gradients -> d(dir_encoding output).  An fp16 two-part product like the hidden layers' since round 3
        //      (KR16 k-steps of v_mfma_f32_32x32x16_f16 x 3; it was (3+C)/2 k-steps of the 64-cycle fp32 MFMA: a fifth of the matrix
        //      time at C = 128).  The operand's exact maximum is known here, so its exponent is picked from it. ----
        const float M_dr = pair_max(array_max(dr));
        // FH: d loss / d g of this lane's sample sits in raw channels 3 + feature (element 16 t + r <-> feature 32 t + rho_h(r)).  It joins
        // G2 where dir_encoding^T reads it (MaskedAddSplitH), but its largest magnitude is needed before that, for the exponent the
        // transient_encoding.0^T / dir_encoding^T pair shares: the values are read TWICE -- here for the maximum only, and again in front of
        // that pair (the second read hits L2) -- rather than kept in 32 registers through the four runs in between (spills; and as the C
        // operand / a VALU update of G2 they broke the fp32 run behind the head's: tools/dbg_fh.py).
        auto load_dg = [&](float (&dst)[FH ? 16 * NTH : 1]) {
            if constexpr (FH) {
                int S_g = a.S;
                asm volatile("" : "+s"(S_g)::"memory");
                if (a.g_gmap) {
                    const float ws = valid ? ld_stream(a.g_raw_t + chan0 + (size_t)3 * S_g) : 0.f;
                    const float* gv = a.g_gmap + (size_t)ray * (a.gout + 1) + 4 * h;          // (one row per ray: the same 260 bytes for a ray's samples)
#pragma unroll
                    for (int e = 0; e < 16 * NTH; ++e) dst[e] = ws * gv[32 * (e >> 4) + nefes_rho(0, e & 15)];
                } else {
                const float* gg = a.g_raw_t + chan0 + (size_t)(3 + 4 * h) * S_g;
#pragma unroll
                for (int e = 0; e < 16 * NTH; ++e) dst[e] = valid ? ld_stream(gg + (size_t)(32 * (e >> 4) + nefes_rho(0, e & 15)) * S_g) : 0.f;
                }
            } else {
                dst[0] = 0.f;
            }
        };
        float M_dg = 0.f;
        if constexpr (FH) {
            float tmp[16 * NTH];
            load_dg(tmp);
            M_dg = pair_max(array_max(tmp));
        }
        int es_g2;
        {
            const int ew = wexp(NEFES_H3B_RGB), tau = tau_of(M_dr, ew);
            es_g2 = tau + ew;
            mma_run_h3<NTH, KR16, 0, true>(ring, ring_lane, ArraySplitH<8 * KR16>{dr, pow2i(tau)}, ZeroInit{}, G2);
        }
        const float M_g2 = rowb(NEFES_H3B_RGB) * M_dr + M_dg;
        float M = 0.f;
        int es3 = 0;
        if constexpr (HAS_T) {
        // ---- transient heads^T (fp32): 5 pre-activation gradients -> d(transient_encoding.4 output), exponent 0 ----
        mma_run<NTH, 3, 0, true>(ring, ring_lane, ArrayIn<3>{dth}, ZeroInit{}, T3);
        M = rowb(NEFES_H3B_TH) * pair_max(array_max(dth));
        // ---- transient_encoding.4^T, .2^T ----
        int es4;
        {
            load_bits(bh, MW_TRUNK + 3 * WH, WH);
            const int ew = wexp(NEFES_H3B_T2), tau = tau_of(M, ew);
            float mx = 0.f;
            es4 = tau + ew;
            mma_run_h3<NTH, GS / 8, 0, true>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTH, WH, 0>{T3, bh, pow2i(tau), mx}, gptr(NEFES_TB_T2), 1.f), ZeroInit{}, T4);
            M = rowb(NEFES_H3B_T2) * pair_max(mx);
        }
        {
            load_bits(bh, MW_TRUNK + 2 * WH, WH);
            const int ew = wexp(NEFES_H3B_T1), tau = tau_of(M, ew);
            float mx = 0.f;
            es3 = tau + ew;
            mma_run_h3<NTH, GS / 8, 0, true>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTH, WH, 0>{T4, bh, pow2i(tau - es4), mx}, gptr(NEFES_TB_T1), pow2i(-es4)), ZeroInit{}, T3);
            M = rowb(NEFES_H3B_T1) * (pair_max(mx) * pow2i(-es4));
        }
        }
        // Full-width accumulators, ping-pong.  Tiles [2, NTW+2) hold a layer's d hidden; XA tile 1 = d dir-embedding;
        // XB tiles 0,1 = d xyz-embedding (written by layer 5, accumulated by layer 1).  In the Wd = 256 objects the 2 x NTW hidden
        // tiles fill the 256 AGPRs (asm MFMAs, gap-by-gap schedule of field_h3.h) and the three embedding tiles live in VGPRs.
        f32x16 XA[NTW + 2], XB[NTW + 2];
        // ---- [transient_encoding.0 ; dir_encoding]^T -> d dir-embedding (tile 1) + d final (tiles 2..): both products
        //      accumulate into the same tiles, so both operands are brought to one common exponent ----
        int es_dt;
        {
            const int ew = wexp(NEFES_H3B_DIR);                                   // = wexp(NEFES_H3B_T0): one scale for the pair (pack.cpp)
            const int tau = tau_of(HAS_T ? fmaxf(M, M_g2) : M_g2, ew);
            float mt = 0.f, mg = 0.f;
            es_dt = tau + ew;
            float dgv[FH ? 16 * NTH : 1];
            load_dg(dgv);                                          // (in flight behind the transient_encoding.0^T run)
            if constexpr (HAS_T) {
                load_bits(bh, MW_TRUNK + WH, WH);
                mma_run_h3<NTW + 1, GS / 8, 1, true>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTH, WH, 0>{T3, bh, pow2i(tau - es3), mt}, gptr(NEFES_TB_T0), pow2i(-es3)), ZeroInit{}, XA);
            }
            load_bits(bh, MW_TRUNK, WH);
#ifdef NEFES_FH_VARIANT_VALU_UPDATE
            // Round 5's failing bring-up form as far as it can be reconstructed, kept buildable for tools/fh_variant.sh (DESIGN.md
            // 4.10): d loss / d g added to the head's tiles by the vector ALU, the pair's second product on the plain functor.  The
            // linter is red on it without the result fences (the transient heads' fp32 tile read 2 wait states behind its MFMA) --
            // and it is numerically RIGHT with and without them on round 6's boxes: round 5's 10-25 % error did not come back.
            if constexpr (FH) {
#pragma unroll
                for (int e = 0; e < 16 * NTH; ++e) G2[e >> 4][e & 15] += dgv[e] * pow2i(es_g2);
                mma_run_h3<NTW + 1, GS / 8, 1, !HAS_T>(ring, ring_lane, MaskedSplitH<NTH, WH, 0>{G2, bh, pow2i(tau - es_g2), mg}, ZeroInit{}, XA);
            } else
#else
            if constexpr (FH)
                mma_run_h3<NTW + 1, GS / 8, 1, !HAS_T>(ring, ring_lane, MaskedAddSplitH<NTH, WH, 0, 16 * NTH>{G2, bh, pow2i(tau - es_g2), mg, dgv, pow2i(es_g2)}, ZeroInit{}, XA);
            else
#endif
            mma_run_h3<NTW + 1, GS / 8, 1, !HAS_T>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTH, WH, 0>{G2, bh, pow2i(tau - es_g2), mg}, gptr(NEFES_TB_DIR), pow2i(-es_g2)), ZeroInit{}, XA);
            M = (HAS_T ? rowb(NEFES_H3B_T0) * (pair_max(mt) * pow2i(-es3)) : 0.f) + rowb(NEFES_H3B_DIR) * (pair_max(mg) * pow2i(-es_g2));
        }
        // The d dir-embedding tile (XA tile 1) is complete here and nothing else reads it: its embedding backward runs NOW and three
        // numbers per lane stay, instead of a 16-register tile riding along through all nine full-width layers (round 3: with the two
        // d xyz-embedding tiles treated the same way below, the layer chain holds exactly the 2 x NTW ping-pong tiles).
        float gv[3];
        {
            float dDv[16], v3[3];
            const float inv = pow2i(-es_dt);
#pragma unroll
            for (int r = 0; r < 16; ++r) dDv[r] = XA[1][r] * inv;
#pragma unroll
            for (int c = 0; c < 3; ++c) v3[c] = STASH(3 + c);
            embed_slots_bwd<NEFES_N_FREQ_DIR>(gv, dDv, v3, h);
            if constexpr (ENC == NEFES_XYZ_HASHGRID_FUSED) {
                // parked in LDS (over the view direction, which is not needed again): left in registers, hipcc sank this whole
                // computation to the end of the tile in this instance -- the tile alive until then, spilled, and accumulator tiles
                // moved inside the asm-scheduled runs (tests/test_pack_stream.py); a store cannot sink past the ring's acquires
#pragma unroll
                for (int c = 0; c < 3; ++c) STASH(3 + c) = gv[c];
            }
        }
        // ---- xyz_encoding_final^T (no ReLU on its output) + static_sigma^T (one extra fp32 k-step) -> d h8 ----
        int es_b;
        {
            const int ew = wexp(NEFES_H3B_FINAL), tau = tau_of(M, ew);
            float mx = 0.f;
            es_b = tau + ew;
            mma_run_h3<NTW, W / 16, 2, true, 2, (WIDE_LAYERS & 1) != 0>(ring, ring_lane, wrap_store_h3<TRAIN>(IdentSplitH<NTW + 2, 2>{XA, pow2i(tau - es_dt), mx}, gptr(NEFES_TB_FINAL), pow2i(-es_dt)), ZeroInit{}, XB);
            const float dsig = STASH(6);
            float dsg[1];
            dsg[0] = dsig * pow2i(es_b);
            // (no trailing fence where layer 8's run is a wide one: its entry fence is the same 18 wait states, and a second
            // scheduling barrier here made hipcc move an accumulator tile inside that run in the hash-grid instance)
#if defined(H3B_WIDE) && defined(H3_WIDE_ENTRY_FENCE)
            mma_run<NTW, 1, 2, false, ((WIDE_LAYERS >> 8) & 1) == 0 || (NTW < NEFES_H3_WIDE_MIN)>(ring, ring_lane, ArrayIn<1>{dsg}, ZeroInit{}, XB);
#else
            mma_run<NTW, 1, 2, false>(ring, ring_lane, ArrayIn<1>{dsg}, ZeroInit{}, XB);
#endif
            M = rowb(NEFES_H3B_FINAL) * (pair_max(mx) * pow2i(-es_dt)) + rowb(NEFES_H3B_SIG) * pair_max(fabsf(dsig));
        }
        // ---- xyz_encoding_8^T .. xyz_encoding_2^T, straight-line (XB -> XA -> XB ...).  Layer 5 also emits the skip's d
        //      xyz-embedding into XB tiles 0,1. ----
        int es_a = 0, es_e = 0;
#define NEFES_BWD_LAYER(L, SRC, DST, ES_SRC, ES_DST, NTILES, T0)                                                       \
        {                                                                                                           \
            load_bits(bt, ((L) - 1) * WT, WT);                                                                      \
            const int ew = wexp(NEFES_H3B_L8 + 8 - (L)), tau = tau_of(M, ew);                                       \
            float mx = 0.f;                                                                                         \
            ES_DST = tau + ew;                                                                                      \
            mma_run_h3<NTILES, W / 16, T0, true, 2, ((WIDE_LAYERS >> (L)) & 1) != 0>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTW + 2, WT, 2>{SRC, bt, pow2i(tau - ES_SRC), mx}, gptr(NEFES_TB_L1 + (L) - 1), pow2i(-(ES_SRC))), ZeroInit{}, DST); \
            M = rowb(NEFES_H3B_L8 + 8 - (L)) * (pair_max(mx) * pow2i(-(ES_SRC)));                                   \
        }
        NEFES_BWD_LAYER(8, XB, XA, es_b, es_a, NTW, 2)
        NEFES_BWD_LAYER(7, XA, XB, es_a, es_b, NTW, 2)
        NEFES_BWD_LAYER(6, XB, XA, es_b, es_a, NTW, 2)
        NEFES_BWD_LAYER(5, XA, XB, es_a, es_b, NTW + 2, 0)
        es_e = es_b;                                                         // exponent of the d xyz-embedding tiles XB[0], XB[1]
        // the skip connection's share of d xyz-embedding (XB tiles 0, 1), consumed at once (see the d dir-embedding tile above)
        float gx[3] = {0.f, 0.f, 0.f};
        float ge[ENC != NEFES_XYZ_FREQ10 ? NEFES_X_STEPS : 1];
        {
            const float inv = pow2i(-es_e);
            if constexpr (ENC != NEFES_XYZ_FREQ10) {        // kept in registers until layer 1's share exists (see part 1's note below)
#pragma unroll
                for (int s_ = 0; s_ < NEFES_X_STEPS; ++s_) ge[s_] = XB[0][s_] * inv;
            } else {
                float dE[32], x3[3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dE[t * 16 + r] = XB[t][r] * inv;
#pragma unroll
                for (int c = 0; c < 3; ++c) x3[c] = STASH(c);
                embed_slots_bwd<NEFES_N_FREQ_XYZ>(gx, dE, x3, h);
                ge[0] = 0.f;
            }
        }
        NEFES_BWD_LAYER(4, XB, XA, es_b, es_a, NTW, 2)
        NEFES_BWD_LAYER(3, XA, XB, es_a, es_b, NTW, 2)
        NEFES_BWD_LAYER(2, XB, XA, es_b, es_a, NTW, 2)
#undef NEFES_BWD_LAYER
        // ---- xyz_encoding_1^T -> layer 1's share of d xyz-embedding, into fresh tiles (the skip's share is already consumed) ----
        int es_1;
        {
            load_bits(bt, 0, WT);
            const int ew = wexp(NEFES_H3B_L1), tau = tau_of(M, ew);
            float mx = 0.f;
            es_1 = tau + ew;
            mma_run_h3<2, W / 16, 0, true>(ring, ring_lane, wrap_store_h3<TRAIN>(MaskedSplitH<NTW + 2, WT, 2>{XA, bt, pow2i(tau - es_a), mx}, gptr(NEFES_TB_L1), pow2i(-es_a)), ZeroInit{}, XB);
        }
        // ---- embedding backward (Embedder.embed :257-267): layer 1's share; the other two were taken where they were produced ----
        const float inv1 = pow2i(-es_1);
        if constexpr (ENC == NEFES_XYZ_HASHGRID_FUSED) {
            // d loss / d pts through the hash grid HERE (frozen table): the [M, 32] encoding gradient (10 GB per frame at 854x480) is
            // neither written nor re-read by a separate gather launch; both lane halves return their eight levels' share
            float x3[3];
#pragma unroll
            for (int s_ = 0; s_ < NEFES_X_STEPS; ++s_) STASH(8 + s_) = ge[s_] + XB[0][s_] * inv1;
#pragma unroll
            for (int c = 0; c < 3; ++c) x3[c] = STASH(c);
            // the position passes through an opaque statement HERE: the cells, indices and table gathers depend on nothing but it, and
            // hipcc otherwise starts them in front of the layer chain (128 registers of gathered entries alive through every run:
            // spills, and accumulator tiles moved inside the asm-scheduled stretches -- tests/test_pack_stream.py caught it)
            asm volatile("" : "+v"(x3[0]), "+v"(x3[1]), "+v"(x3[2]));
            hg_encode_slots_bwd(gx, [&](int s_) { return STASH(8 + s_); }, x3, h, *hg_lds, a.hg_table);
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = STASH(3 + c);
        } else if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
#pragma unroll
            for (int s_ = 0; s_ < NEFES_X_STEPS; ++s_) ge[s_] += XB[0][s_] * inv1;
            if (valid) {
                float* gp = a.g_enc + m * 32 + h;
#pragma unroll
                for (int s_ = 0; s_ < NEFES_X_STEPS; ++s_) gp[2 * s_] = ge[s_];
            }
        } else {
            float dE[32], x3[3], g1[3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dE[t * 16 + r] = XB[t][r] * inv1;
#pragma unroll
            for (int c = 0; c < 3; ++c) x3[c] = STASH(c);
            embed_slots_bwd<NEFES_N_FREQ_XYZ>(g1, dE, x3, h);
#pragma unroll
            for (int c = 0; c < 3; ++c) gx[c] += g1[c];
        }
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

template <int W, int KR16, int ENC, bool HAS_T = true, bool TRAIN = false, bool FH = false>
static int launch_bwd_h3(const FieldBwdH3Args& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_H3B_SLOTS * NEFES_SLAB_BYTES
                       + (size_t)4 * (8 * (W / 64) + 4 * (W / 128) + (ENC == NEFES_XYZ_HASHGRID_FUSED ? 24 : 8)) * 256 + 256
                       + (ENC == NEFES_XYZ_HASHGRID_FUSED ? 512 : 0);
    static_assert(sizeof(HgGeom) <= 512 && 2 * NEFES_H3B_N <= 64, "scale table (256 bytes) + the level geometry's LDS slot");
    auto k = field_bwd_h3_kernel<W, KR16, ENC, HAS_T, TRAIN, FH>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int slots = cus * NEFES_H3B_WG_PER_CU(W);            // Wd = 128: two workgroups share a CU (61 KiB of LDS each)
    int grid = a.n_tiles < slots ? a.n_tiles : slots;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

// Instances spread over nine objects built from this one source (Makefile: -DNEFES_TU_PART=0..8; even parts from 2 on are the
// Wd = 128 objects): part 0 = entry points + Wd = 256 / head class 0, part 1 = hash-grid instance, part 2 = Wd = 128 / class 1,
// parts 3 / 4 = their TRAIN instances, parts 5 / 6 = Wd = 256 / class 1 and Wd = 128 / class 0, parts 7 / 8 = their TRAIN instances.
#ifndef NEFES_TU_PART
#define NEFES_TU_PART 0
#endif
enum { BWD_H3_EXT = 0, BWD_H3_FULL, BWD_H3_TRAIN_STATIC, BWD_H3_TRAIN_FULL, BWD_H3_STATIC, BWD_H3_HG, BWD_H3_FH };
int nefes_bwd_h3_launch_part1(int which, const FieldBwdH3Args& a, hipStream_t st);
int nefes_bwd_h3_launch_part2(int which, const FieldBwdH3Args& a, hipStream_t st);
int nefes_bwd_h3_launch_part3(int which, const FieldBwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 256, class 0
int nefes_bwd_h3_launch_part4(int which, const FieldBwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 128, class 1
int nefes_bwd_h3_launch_part5(int which, const FieldBwdH3Args& a, hipStream_t st);   // Wd = 256, class 1
int nefes_bwd_h3_launch_part6(int which, const FieldBwdH3Args& a, hipStream_t st);   // Wd = 128, class 0
int nefes_bwd_h3_launch_part7(int which, const FieldBwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 256, class 1
int nefes_bwd_h3_launch_part8(int which, const FieldBwdH3Args& a, hipStream_t st);   // TRAIN instances, Wd = 128, class 0

#if NEFES_TU_PART == 1
int nefes_bwd_h3_launch_part1(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_EXT) return launch_bwd_h3<256, 2, NEFES_XYZ_EXTERNAL32>(a, st);
    if (which == BWD_H3_STATIC) return launch_bwd_h3<256, 2, NEFES_XYZ_FREQ10, false>(a, st);     // static head alone, inference (round 5)
    // The hash grid's backward in the epilogue (round 5), on the gap-by-gap schedule like its neighbours.  Its first form parked the
    // skip connection's share of d encoding in LDS across layers 4..1 and hipcc then split an accumulator tile's live range inside
    // an asm-scheduled run (a v_accvgpr_mov behind an MFMA that has not written the tile yet: tests/test_pack_stream.py caught it,
    // no numerical test did); with the share in sixteen registers, as the external-encoding instance keeps it, the tiles stay put.
    if (which == BWD_H3_HG) return launch_bwd_h3<256, 2, NEFES_XYZ_HASHGRID_FUSED>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 2      // built with -mllvm -amdgpu-mfma-vgpr-form: see field_fwd_h3.hip
int nefes_bwd_h3_launch_part2(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_FULL) return launch_bwd_h3<128, 9, NEFES_XYZ_FREQ10>(a, st);
    if (which == BWD_H3_STATIC) return launch_bwd_h3<128, 9, NEFES_XYZ_FREQ10, false>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 3
int nefes_bwd_h3_launch_part3(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_TRAIN_STATIC) return launch_bwd_h3<256, 2, NEFES_XYZ_FREQ10, false, true>(a, st);
    if (which == BWD_H3_TRAIN_FULL) return launch_bwd_h3<256, 2, NEFES_XYZ_FREQ10, true, true>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 4      // (built like part 2)
int nefes_bwd_h3_launch_part4(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_TRAIN_STATIC) return launch_bwd_h3<128, 9, NEFES_XYZ_FREQ10, false, true>(a, st);
    if (which == BWD_H3_TRAIN_FULL) return launch_bwd_h3<128, 9, NEFES_XYZ_FREQ10, true, true>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 5      // (gap-by-gap schedule, like part 0)
int nefes_bwd_h3_launch_part5(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_FULL) return launch_bwd_h3<256, 9, NEFES_XYZ_FREQ10>(a, st);
    if (which == BWD_H3_STATIC) return launch_bwd_h3<256, 9, NEFES_XYZ_FREQ10, false>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 6      // (built like part 2)
int nefes_bwd_h3_launch_part6(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_FULL) return launch_bwd_h3<128, 2, NEFES_XYZ_FREQ10>(a, st);
    if (which == BWD_H3_STATIC) return launch_bwd_h3<128, 2, NEFES_XYZ_FREQ10, false>(a, st);
    if (which == BWD_H3_FH) return launch_bwd_h3<128, 2, NEFES_XYZ_FREQ10, true, false, true>(a, st);      // factored head (round 5)
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 7
int nefes_bwd_h3_launch_part7(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_TRAIN_STATIC) return launch_bwd_h3<256, 9, NEFES_XYZ_FREQ10, false, true>(a, st);
    if (which == BWD_H3_TRAIN_FULL) return launch_bwd_h3<256, 9, NEFES_XYZ_FREQ10, true, true>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#elif NEFES_TU_PART == 8      // (built like part 2)
int nefes_bwd_h3_launch_part8(int which, const FieldBwdH3Args& a, hipStream_t st) {
    if (which == BWD_H3_TRAIN_STATIC) return launch_bwd_h3<128, 2, NEFES_XYZ_FREQ10, false, true>(a, st);
    if (which == BWD_H3_TRAIN_FULL) return launch_bwd_h3<128, 2, NEFES_XYZ_FREQ10, true, true>(a, st);
    return NEFES_E_UNSUPPORTED;
}
#else   // part 0

// The fused dX chain of the train-mode backward on the fp16 pipe: as nefes_field_bwd_train (field_bwd.hip), same `dacts` rows.
extern "C" int nefes_field_bwd_train_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                        const float* rays_d, const float* z, const float* viewdirs, const float* raw_t,
                                        const float* g_raw_t, const uint32_t* masks, float* dacts, float* g_pts,
                                        float* g_viewdirs_s, void* stream) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !dacts || !g_pts || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    if (!(rays_o && rays_d && z)) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_STATIC && mode != NEFES_FIELD_FULL) return NEFES_E_BADARG;
    const bool full = mode == NEFES_FIELD_FULL;
    if (full && !desc->has_transient) return NEFES_E_UNSUPPORTED;
    const int cls = nefes_head_class(desc->feat_dim);
    if ((desc->width != 256 && desc->width != 128) || cls < 0 || desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[full ? NEFES_STREAM_BWD_FULL_H3 : NEFES_STREAM_BWD_STATIC_H3];
    if (si.n_slabs == 0 || si.scale_count < 2 * (uint32_t)(full ? NEFES_H3B_N : NEFES_H3B_N_STATIC)) return NEFES_E_UNSUPPORTED;
    FieldBwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.tab = (const int*)((const char*)packed + si.bias_off) + si.scale_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = nullptr; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_enc = nullptr; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + (full ? 6 : 1);
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    a.dacts = dacts; a.gout = 0; a.hg_table = nullptr; a.g_gmap = nullptr;
    a.rows = nefes_train_row(desc->width, desc->feat_dim, NEFES_TB_END);
    const int which = full ? BWD_H3_TRAIN_FULL : BWD_H3_TRAIN_STATIC;
    hipStream_t st = (hipStream_t)stream;
    if (desc->width == 256) return cls == 0 ? nefes_bwd_h3_launch_part3(which, a, st) : nefes_bwd_h3_launch_part7(which, a, st);
    return cls == 1 ? nefes_bwd_h3_launch_part4(which, a, st) : nefes_bwd_h3_launch_part8(which, a, st);
}

// nefes_field_bwd_static on the fp16 pipe: backward-to-inputs of a NEFES_FIELD_STATIC forward with frozen weights (round 5: every
// compiled (width, head class) pair; the fp32-MFMA nefes_field_bwd_static serves the two canonical shapes only).
extern "C" int nefes_field_bwd_static_h3(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                         const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                         const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                         float* g_viewdirs_s, void* stream) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !g_pts || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    if (!pts && !(rays_o && rays_d && z)) return NEFES_E_BADARG;
    const int cls = nefes_head_class(desc->feat_dim);
    if ((desc->width != 256 && desc->width != 128) || cls < 0 || desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[NEFES_STREAM_BWD_STATIC_H3];
    if (si.n_slabs == 0 || si.scale_count < 2 * (uint32_t)NEFES_H3B_N_STATIC) return NEFES_E_UNSUPPORTED;
    FieldBwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.tab = (const int*)((const char*)packed + si.bias_off) + si.scale_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_enc = nullptr; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + 1;
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    a.dacts = nullptr; a.rows = 0; a.gout = 0; a.hg_table = nullptr; a.g_gmap = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (desc->width == 256) return cls == 0 ? nefes_bwd_h3_launch_part1(BWD_H3_STATIC, a, st) : nefes_bwd_h3_launch_part5(BWD_H3_STATIC, a, st);
    return cls == 1 ? nefes_bwd_h3_launch_part2(BWD_H3_STATIC, a, st) : nefes_bwd_h3_launch_part6(BWD_H3_STATIC, a, st);
}

static int field_bwd_h3_impl(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                  const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                  float* g_xyz_enc, float* g_viewdirs_s, void* stream, const NefesHashGridDesc* grid, const float* table,
                                  bool fh = false, const float* g_gmap = nullptr) {
    if (!desc || !packed || !viewdirs || !raw_t || !g_raw_t || !masks || !g_viewdirs_s || N <= 0 || S <= 0)
        return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    const bool fused_grid = ext && table != nullptr;            // d pts through the hash grid inside the kernel (hashgrid.h)
    if (fused_grid ? !(g_pts && grid && rays_o && rays_d && z) : (ext ? !g_xyz_enc : (!g_pts || (!pts && !(rays_o && rays_d && z))))) return NEFES_E_BADARG;
    if (!desc->has_transient) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[NEFES_STREAM_BWD_FULL_H3];
    if (si.n_slabs == 0 || si.scale_count < 2 * NEFES_H3B_N) return NEFES_E_UNSUPPORTED;
    FieldBwdH3Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.tab = (const int*)((const char*)packed + si.bias_off) + si.scale_off;
    a.n_slabs = si.n_slabs;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs;
    a.raw_t = raw_t; a.g_raw_t = g_raw_t; a.masks = masks; a.g_pts = g_pts; a.g_enc = g_xyz_enc; a.g_vs = g_viewdirs_s;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = 3 + a.C + 6;
    a.gout = 0; a.g_gmap = g_gmap;
    if (fh) {                                                     // factored head: see nefes_field_fwd_h3_fh
        if (desc->feat_dim != 0 || desc->width != 128 || ext) return NEFES_E_UNSUPPORTED;
        a.gout = desc->width / 2;
        a.C = a.gout + 1;
        a.R = 3 + a.C + 6;
    }
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    a.dacts = nullptr; a.rows = 0;
    a.hg_table = (const float2*)table;
    if (fused_grid) {
        rc = hg_geometry(grid, &a.hg, nullptr);
        if (rc) return rc;
        if (a.hg.n_levels != 16) return NEFES_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int cls = nefes_head_class(desc->feat_dim);          // compiled set: as nefes_field_fwd_h3
    if (cls < 0) return NEFES_E_UNSUPPORTED;
    if (fh) return nefes_bwd_h3_launch_part6(BWD_H3_FH, a, st);
    if (desc->width == 256 && fused_grid) return cls == 0 ? nefes_bwd_h3_launch_part1(BWD_H3_HG, a, st) : NEFES_E_UNSUPPORTED;
    if (desc->width == 256 && ext) return cls == 0 ? nefes_bwd_h3_launch_part1(BWD_H3_EXT, a, st) : NEFES_E_UNSUPPORTED;
    if (desc->width == 256) return cls == 0 ? launch_bwd_h3<256, 2, NEFES_XYZ_FREQ10>(a, st) : nefes_bwd_h3_launch_part5(BWD_H3_FULL, a, st);
    if (desc->width == 128 && !ext) return cls == 1 ? nefes_bwd_h3_launch_part2(BWD_H3_FULL, a, st) : nefes_bwd_h3_launch_part6(BWD_H3_FULL, a, st);
    return NEFES_E_UNSUPPORTED;
}

extern "C" int nefes_field_bwd_h3(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* viewdirs,
                                  const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts,
                                  float* g_xyz_enc, float* g_viewdirs_s, void* stream) {
    return field_bwd_h3_impl(desc, packed, N, S, rays_o, rays_d, z, pts, viewdirs, raw_t, g_raw_t, masks, g_pts, g_xyz_enc, g_viewdirs_s, stream,
                             nullptr, nullptr);
}

// nefes_field_bwd_h3 for a nefes_field_fwd_h3_fh forward: g_raw_t [N][3 + (W/2 + 1) + 6][S] carries d loss / d g in channels 3 .. 3 + W/2
// (the gradient of the ones channel is ignored).
extern "C" int nefes_field_bwd_h3_fh(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o, const float* rays_d,
                                     const float* z, const float* viewdirs, const float* raw_t, const float* g_raw_t, const float* g_gmap,
                                     const uint32_t* masks, float* g_pts, float* g_viewdirs_s, void* stream) {
    if (!rays_o || !rays_d || !z) return NEFES_E_BADARG;
    return field_bwd_h3_impl(desc, packed, N, S, rays_o, rays_d, z, nullptr, viewdirs, raw_t, g_raw_t, masks, g_pts, nullptr, g_viewdirs_s, stream,
                             nullptr, nullptr, true, g_gmap);
}

// nefes_field_bwd_h3 for a nefes_field_fwd_h3_hashgrid forward: g_pts [N*S, 3] = d loss / d (o + d z) through the MLP AND the hash
// grid (frozen table) -- what nefes_field_bwd_h3 (g_xyz_enc) followed by nefes_hashgrid_bwd_x compute, without the [M, 32] gradient.
extern "C" int nefes_field_bwd_h3_hashgrid(const NefesNetDesc* desc, const void* packed, const NefesHashGridDesc* grid,
                                           const float* table, int N, int S, const float* rays_o, const float* rays_d, const float* z,
                                           const float* viewdirs, const float* raw_t, const float* g_raw_t, const uint32_t* masks,
                                           float* g_pts, float* g_viewdirs_s, void* stream) {
    if (!desc || desc->xyz_encoding != NEFES_XYZ_EXTERNAL32 || !grid || !table) return NEFES_E_BADARG;
    return field_bwd_h3_impl(desc, packed, N, S, rays_o, rays_d, z, nullptr, viewdirs, raw_t, g_raw_t, masks, g_pts, nullptr, g_viewdirs_s, stream,
                             grid, table);
}
#endif   // NEFES_TU_PART
