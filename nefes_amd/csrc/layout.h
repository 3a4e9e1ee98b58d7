// Register/stream layout shared by the host-side weight packer (pack.cpp) and the
// fused field kernels (field_fwd.hip / field_bwd.hip).
//
// The field MLP (reference: script/models/nerfh_nff.py:421-576) is evaluated
// TRANSPOSED:  H'^T[out, sample] = W[out, in] * H^T[in, sample]  with
// v_mfma_f32_32x32x2_f32 (A = weights, B = activations).  With that orientation the
// 32x32 accumulator tile of one layer (lane = sample column, registers = feature
// rows) is, register for register, the B operand of the next layer, so
// activations never leave the register file.  Per wave: 32 samples.
//
// Vocabulary
//   slot (s, h)  : register index s of an activation vector, lane half h = lane>>5.
//                  One MFMA k-step consumes register s: half 0 supplies k = slot(s,0),
//                  half 1 supplies k = slot(s,1).
//   tile (t, i)  : accumulator tile t, row i (0..31).  Row i lives in register
//                  r = (i&3) + 4*(i>>3) of lane half h = (i>>2)&1   [CDNA C/D map].
//   fragment     : 64 lanes x 1 float = 256 B; the A operand of one MFMA.
//   slab         : 32 KiB = up to 128 fragments in consumption order; the unit of
//                  the LDS-DMA weight ring.  A slab never spans two segments.
//   segment      : one (activation vector) x (weight block) product accumulated into
//                  NT tiles over KS k-steps.
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#define NEFES_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define NEFES_HD inline
#endif

// Slab sizes (one workgroup barrier per slab): 32 KiB measured +0.7..1 % over 16 KiB (half the barriers).  Forward and
// backward streams are sized separately; a kernel translation unit defines NEFES_SLAB_KIB (to one of the two) before
// including field_common.h; the packer knows both.
#ifndef NEFES_FWD_SLAB_KIB
#define NEFES_FWD_SLAB_KIB 32
#endif
#ifndef NEFES_BWD_SLAB_KIB
#define NEFES_BWD_SLAB_KIB 32
#endif
#define NEFES_FRAGS_OF_KIB(kib) ((kib) * 4)               /* 256-byte fragments per slab */
#define NEFES_SLAB_BYTES (NEFES_SLAB_KIB * 1024)          /* (expand where used: the kernel TU's NEFES_SLAB_KIB) */
#define NEFES_SLAB_FRAGS NEFES_FRAGS_OF_KIB(NEFES_SLAB_KIB)
#define NEFES_SLAB_PIECES (NEFES_SLAB_BYTES / 4096)       /* 1 KiB LDS-DMA pieces per wave and slab */
#define NEFES_RING_SLOTS (128 / NEFES_FWD_SLAB_KIB)       /* forward ring: 128 KiB */
#define NEFES_BWD_SLOTS (96 / NEFES_BWD_SLAB_KIB)         /* backward ring: 96 KiB (+ the tile's ReLU masks, <= 40 KiB) */
#define NEFES_N_FREQ_XYZ 10
#define NEFES_N_FREQ_DIR 4
#define NEFES_E_STEPS 32   /* 63 xyz-embedding features + 1 pad, two per k-step */
#define NEFES_D_STEPS 14   /* 27 dir-embedding features + 1 pad */
#define NEFES_X_STEPS 16   /* 32 features of an externally supplied xyz embedding (hash grid), compact slots 2s+h */

// accumulator row of (register r, lane half h)
NEFES_HD int nefes_rho(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// inverse: row i -> register / half
NEFES_HD int nefes_row_reg(int i) { return (i & 3) + 4 * (i >> 3); }
NEFES_HD int nefes_row_half(int i) { return (i >> 2) & 1; }
// natural hidden-vector slot -> feature index
NEFES_HD int nefes_nat_slot(int s, int h) { return 32 * (s >> 4) + nefes_rho(h, s & 15); }
// frequency-embedding slot -> index in the reference's embedding order
//   reference order (nerfh_nff.py:257-267): [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...]
//   ours: step s < 3L holds (sin | cos) of frequency s/3, axis s%3 in (half 0 | half 1);
//         step 3L holds (x0 | x1); step 3L+1 holds (x2 | pad).   -1 = pad.
NEFES_HD int nefes_emb_slot(int L, int s, int h) {
    if (s < 3 * L) return 3 + 6 * (s / 3) + 3 * h + (s % 3);
    if (s == 3 * L) return h;
    if (s == 3 * L + 1) return h == 0 ? 2 : -1;
    return -1;
}
// Head classes.  The static rgb+feature head (3 + C output rows, nerfh_nff.py:487-490; C = FEATURE_DIM, :21) is a compile-time
// shape in the fp16 two-part field kernels: NTR output tiles forward, KR16 k-steps of 16 upstream channels backward.  Two classes
// are compiled per width and the packer pads a network's head to its class (zero rows / zero columns), so that C is a run-time
// parameter: class 0 serves 3 + C <= 32 (C = 16: BASELINE configs[1]), class 1 serves 3 + C <= 144 (C = 128: the reference's own
// FEATURE_DIM).  -1: no instance.
NEFES_HD int nefes_head_class(int C) { return C < 0 ? -1 : (3 + C <= 32 ? 0 : (3 + C <= 144 ? 1 : -1)); }
NEFES_HD int nefes_head_ntr(int C) { return 3 + C <= 32 ? 1 : 5; }
NEFES_HD int nefes_head_kr16(int C) { return 3 + C <= 32 ? 2 : 9; }
#define NEFES_HEAD_MAX_C 141
// k-steps per slab for a segment with NT accumulator tiles
NEFES_HD int nefes_steps_per_slab(int nt, int slab_frags) { return slab_frags / nt; }
NEFES_HD int nefes_segment_slabs(int nt, int ks, int slab_frags) {
    const int sps = slab_frags / nt;
    return (ks + sps - 1) / sps;
}

// stream kinds inside a packed blob
// _X6 streams: trunk layers 2..8 as bf16x6 split products (v_mfma_f32_32x32x16_bf16 on exact hi/mid/lo bf16 triples, six
// cross terms: fp32-level accuracy), everything else as in the fp32 stream; 48 KiB slabs = 16 units of 3 KiB.
// _H3 streams: every product of the forward pass and the hidden products of the backward pass as fp16 two-part split
// products (v_mfma_f32_32x32x16_f16 on (hi, lo) fp16 pairs of power-of-two scaled operands, three cross terms hh + hl + lh:
// fp32-level accuracy at half the matrix-core work of bf16x6; field_h3.h); units of 2 KiB = hi | lo A operands; the
// per-segment weight-scale exponents ride behind the bias blocks (NefesStreamInfo.scale_off).
enum { NEFES_STREAM_FWD_SIGMA = 0, NEFES_STREAM_FWD_STATIC = 1, NEFES_STREAM_FWD_FULL = 2, NEFES_STREAM_BWD_FULL = 3,
       NEFES_STREAM_FWD_SIGMA_X6 = 4, NEFES_STREAM_FWD_FULL_X6 = 5, NEFES_STREAM_BWD_FULL_X6 = 6, NEFES_STREAM_BWD_STATIC = 7,
       NEFES_STREAM_FWD_SIGMA_H3 = 8, NEFES_STREAM_FWD_FULL_H3 = 9, NEFES_STREAM_BWD_FULL_H3 = 10,
       NEFES_STREAM_FWD_STATIC_H3 = 11, NEFES_STREAM_BWD_STATIC_H3 = 12,   /* static head only (coarse network in train mode) */
       NEFES_N_STREAMS = 13 };
#define NEFES_X6_SLAB_KIB 48
#define NEFES_H3_FWD_SLAB_KIB 32       /* 16 units of 2 KiB (3 slots = 96 KiB: the embedding is parked in LDS beside the ring) */
#define NEFES_H3_FWD_SLAB_KIB_128 16   /* Wd = 128 forward streams: 8 units; 2 slots = 32 KiB so that TWO workgroups share a CU's 160 KiB LDS */
#define NEFES_H3_BWD_SLAB_KIB 32       /* 16 units */
#define NEFES_H3_BWD_SLAB_KIB_128 16   /* Wd = 128 backward stream: 8 units (16 staging registers less in a kernel that spills into AGPRs) */
#define NEFES_H3_TARGET_EXP 14         /* operands are scaled so that the largest magnitude lies in [2^14, 2^15) */
NEFES_HD int nefes_stream_slab_kib(int stream, int width) {
    if (stream == NEFES_STREAM_BWD_FULL || stream == NEFES_STREAM_BWD_FULL_X6 || stream == NEFES_STREAM_BWD_STATIC) return NEFES_BWD_SLAB_KIB;
    if (stream == NEFES_STREAM_BWD_FULL_H3 || stream == NEFES_STREAM_BWD_STATIC_H3) return width == 128 ? NEFES_H3_BWD_SLAB_KIB_128 : NEFES_H3_BWD_SLAB_KIB;
    if (stream == NEFES_STREAM_FWD_SIGMA_H3 || stream == NEFES_STREAM_FWD_FULL_H3 || stream == NEFES_STREAM_FWD_STATIC_H3)
        return width == 128 ? NEFES_H3_FWD_SLAB_KIB_128 : NEFES_H3_FWD_SLAB_KIB;
    return stream >= NEFES_STREAM_FWD_SIGMA_X6 ? NEFES_X6_SLAB_KIB : NEFES_FWD_SLAB_KIB;
}
#define NEFES_PACK_KEEP 0xffffffffu   /* nefes_pack_map code: this 32-bit word is not written by the slot expansion (both halves carry it) */
#define NEFES_BLOB_HEADER_BYTES 512   /* NefesBlobInfo, padded: written by the host packer only (pack.cpp, pack_device.hip) */

// Segment ordinals of the _H3 streams.  The stream's scale table (pack.cpp; words behind the bias blocks, NefesStreamInfo.
// scale_off) holds per segment s: word 2s = weight-scale exponent e (int32: the weights are stored times 2^e; 0 for fp32
// segments), word 2s+1 = row bound (float: max over the segment's output rows of the sum of |w| over its k-values, so that
// |W x|_inf <= bound |x|_inf); then, per bias block of the stream in order, max |b| (float).
enum { NEFES_H3F_L1 = 0, NEFES_H3F_L2, NEFES_H3F_L3, NEFES_H3F_L4, NEFES_H3F_L5H, NEFES_H3F_L5E, NEFES_H3F_L6, NEFES_H3F_L7,
       NEFES_H3F_L8, NEFES_H3F_SIG, NEFES_H3F_FINAL, NEFES_H3F_DT_H, NEFES_H3F_DT_D, NEFES_H3F_RGB, NEFES_H3F_T1, NEFES_H3F_T2,
       NEFES_H3F_TH, NEFES_H3F_N };
NEFES_HD int nefes_h3_tab_exp(int seg) { return 2 * seg; }
NEFES_HD int nefes_h3_tab_bound(int seg) { return 2 * seg + 1; }
NEFES_HD int nefes_h3_tab_bias(int n_segs, int block) { return 2 * n_segs + block; }
/* bias blocks of the forward streams, in stream order */
enum { NEFES_H3BB_L1 = 0, NEFES_H3BB_SIG = 8, NEFES_H3BB_FINAL, NEFES_H3BB_DIR, NEFES_H3BB_RGB, NEFES_H3BB_T0, NEFES_H3BB_T1,
       NEFES_H3BB_T2, NEFES_H3BB_TH };
enum { NEFES_H3B_RGB = 0, NEFES_H3B_TH, NEFES_H3B_T2, NEFES_H3B_T1, NEFES_H3B_T0, NEFES_H3B_DIR, NEFES_H3B_FINAL, NEFES_H3B_SIG,
       NEFES_H3B_L8, NEFES_H3B_L7, NEFES_H3B_L6, NEFES_H3B_L5, NEFES_H3B_L4, NEFES_H3B_L3, NEFES_H3B_L2, NEFES_H3B_L1,
       NEFES_H3B_N };
/* the static-head streams hold a subset of those segments in the same order: forward = the first NEFES_H3F_RGB + 1 (DT_H / DT_D
 * are then dir_encoding alone), backward = RGB, then DIR .. L1 (no TH, T2, T1, T0) */
#define NEFES_H3F_N_STATIC (NEFES_H3F_RGB + 1)
#define NEFES_H3B_N_STATIC (NEFES_H3B_N - 4)
NEFES_HD int nefes_h3b_seg(int has_transient, int seg) { return (has_transient || seg == NEFES_H3B_RGB) ? seg : seg - 4; }

// ReLU-mask words (32 bit) written per lane per 32-sample tile by the full forward pass:
// 8 trunk layers (W/64 words each) + dir + 3 transient layers (W/128 words each)
NEFES_HD int nefes_mask_words(int W) { return 8 * (W / 64) + 4 * (W / 128); }

// ---- train-mode activation / gradient buffers (field_fwd TRAIN instances, train.hip) ----
// Tile-major: buf[tile128][rows x 128 samples] fp32; inside a tile, blocks of 32 rows x 16 samples (2 KiB) are contiguous:
//     element (row, sample) at float offset nefes_train_off(row, sample) = [row / 32][sample / 16][row % 32][sample % 16]
// A 16-sample step of the weight-gradient kernels (train.hip) then reads whole contiguous 2 KiB blocks -- with plain
// [row][128] rows every 128-byte line was shared by two steps issued a microsecond apart, and the HBM counters showed 1.4 x the
// algorithmic bytes (the in-flight data of a launch is as large as the L2s).  Row blocks start at multiples of 32.
// One row map for both buffers:
//   `acts`  (forward):  E, DV = embeddings in slot order (row 2s+h); L1..L8, FINAL, DIR, T0..T2 = PRE-activations
//   `dacts` (backward): L1..T2 = gradient w.r.t. those pre-activations; RGB, SIG, TH = head pre-activation gradients
// Hidden blocks are in natural feature order.  Head blocks are padded to whole 32-row tiles.
enum { NEFES_TB_E = 0, NEFES_TB_DV = 1, NEFES_TB_L1 = 2 /* .. L8 = 9 */, NEFES_TB_FINAL = 10, NEFES_TB_DIR = 11,
       NEFES_TB_T0 = 12, NEFES_TB_T1 = 13, NEFES_TB_T2 = 14, NEFES_TB_RGB = 15, NEFES_TB_SIG = 16, NEFES_TB_TH = 17,
       NEFES_TB_END = 18 };
NEFES_HD size_t nefes_train_off(int row, int sample) {
    return ((size_t)(row >> 5) * 8 + (size_t)(sample >> 4)) * 512 + (size_t)(row & 31) * 16 + (size_t)(sample & 15);
}
// the per-lane part of that offset for the kernels whose lane (j = lane % 32, h = lane / 32) of wave w holds sample 32 w + j and
// rows 32 t + rho(0, r) + 4 h of a block: offset = (row0 / 32 + t) * 4096 + rho(0, r) * 16 + nefes_train_lane_off(w, j, h)
NEFES_HD uint32_t nefes_train_lane_off(int wave, int j, int h) { return (uint32_t)(((wave * 32 + j) >> 4) * 512 + 4 * h * 16 + (j & 15)); }
NEFES_HD int nefes_train_row(int W, int C, int block) {
    const int ntr = nefes_head_ntr(C);
    int r = 0;
    if (block == NEFES_TB_E) return r;
    r += 2 * NEFES_E_STEPS;
    if (block == NEFES_TB_DV) return r;
    r += 32;
    if (block >= NEFES_TB_L1 && block <= NEFES_TB_FINAL) return r + (block - NEFES_TB_L1) * W;
    r += 9 * W;
    if (block >= NEFES_TB_DIR && block <= NEFES_TB_T2) return r + (block - NEFES_TB_DIR) * (W / 2);
    r += 4 * (W / 2);
    if (block == NEFES_TB_RGB) return r;
    r += 32 * ntr;
    if (block == NEFES_TB_SIG) return r;
    r += 32;
    if (block == NEFES_TB_TH) return r;
    return r + 32;   // NEFES_TB_END = rows per tile
}
