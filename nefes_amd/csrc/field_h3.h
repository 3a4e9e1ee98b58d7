// fp16 two-part split products on v_mfma_f32_32x32x16_f16 (shared by field_fwd_h3.hip and field_bwd_h3.hip; include after
// field_common.h and field_x6.h, with NEFES_SLAB_KIB defined by the translation unit).
//
//     x 2^ex = xh + xl,  w 2^ew = wh + wl      (two fp16 each: 11 + 11 = 22 significant bits, round-to-nearest parts)
//     w x 2^(ex+ew) ~= wh xh + wh xl + wl xh     (dropped term wl xl: relative size 2^-22)
//
// accumulated in fp32 by the matrix core: fp32-level accuracy (tools/fp16x3_accuracy.py; tests/test_gpu_h3.py against float64)
// at 3 x 32 = 96 cycles per 16 k-values and tile -- half the matrix-core work of bf16x6 (field_x6.h).
//
// fp16 carries 5 exponent bits, so every operand is scaled by a power of two first:
//   * weights per MATRIX, by the packer (pack.cpp: max |w| 2^ew in [2^14, 2^15); ew rides in the stream's exponent table);
//   * activations / gradient vectors per SAMPLE and product, inside the conversion (v_fma_mixlo/mixhi_f16: hi = RNE_f16(x 2^ex),
//     lo = RNE_f16(x 2^ex - hi): four VALU per pair of values).  ex comes from an upper BOUND of the sample's largest magnitude:
//     |W x + b|_inf <= rowbound(W) |x|_inf + |b|_inf, with rowbound and max|b| from the packer's table and |x|_inf measured
//     exactly while x was split for the previous product (one v_max3_f32 per pair of values, on registers the split reads
//     anyway).  Scanning the finished accumulators for their maximum instead costs a read of all 128 accumulator registers per
//     layer and a serial tail: measured 17 % (forward) / 24 % (backward) of the kernel time.  The bound is loose by the ratio
//     rowbound / actual gain (~2^3..2^4 for random weights, not cumulative), i.e. operands land 3-4 binades below 2^15; fp16
//     subnormals (consumed by the MFMA: tools/probe/h3_probe.hip) keep the absolute resolution at 2^-24, 2^-35 of the maximum.
// Powers of two commute with the products, with ReLU and with the sign bits the masks record, so an accumulator set simply
// carries its exponent: acc = 2^es * (true value), es per lane (an int the kernels thread through the layer chain).  Bias tiles enter multiplied by 2^es; products
// that accumulate into the same tiles use a common exponent; raw outputs are multiplied by 2^-es on the way out.
// Values more than 2^-25 below the lane's maximum lose low bits (fp16 subnormals resolve 2^-24): an absolute error of 2^-39
// relative to that maximum, i.e. nothing.
#pragma once
#ifndef NEFES_H3_WIDE_MIN
#define NEFES_H3_WIDE_MIN 8   /* segments with at least this many tiles take the gap-by-gap schedule (mma_run_h3_wide) */
#endif
#ifndef NEFES_H3_WIDE_MAX
#define NEFES_H3_WIDE_MAX 64
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Split2 {
    u32x4 h, l;              // 8 fp16 each: element i in the low/high half of dword i/2
};
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) {
    f16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
__device__ __forceinline__ f16x8 as_f16x8(f32x4 v) {
    f16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// ---- exponent bookkeeping (all per lane, integers: exact) -----------------------------------------------------------------
__device__ __forceinline__ float pow2i(int e) {            // 2^e, e clamped to [-126, 127]
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __uint_as_float((uint32_t)(e + 127) << 23);
}
// exponent ex with m 2^ex in [2^14, 2^15) for the lane pair's largest magnitude m >= 0; 0 when m is zero / tiny
__device__ __forceinline__ int pick_exp(float m) {
    const int be = (int)(__float_as_uint(m) >> 23);        // biased exponent (m >= 0)
    const int ex = NEFES_H3_TARGET_EXP + 127 - be;
    return (be < 20 || be > 250) ? 0 : ex;                 // m < 2^-107 (or inf/nan): leave unscaled
}
// max over the two lanes (j, j + 32) that share a sample, m >= 0.  v_permlane32_swap_b32 of a register with itself returns the
// lower 32 lanes' values on both halves and the upper 32 lanes' values on both halves: no address register, no LDS round trip
// (a ds_bpermute needs the lane^32 byte address alive for the whole kernel).
__device__ __forceinline__ float pair_max(float m) {
    const uint32_t u = __float_as_uint(m);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const uint32_t a = r[0], b = r[1];
    return __uint_as_float(a > b ? a : b);                  // non-negative floats order like their bit patterns
}
// running maximum of the values a product consumes (they sit in VGPRs for the split): one instruction per pair
__device__ __forceinline__ void max3_acc(float& m, float x0, float x1) {
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x0), "v"(x1));
}
__device__ __forceinline__ void absmax3_acc(float& m, float x0, float x1) {
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(x0), "v"(x1));
}
template <int N>
__device__ __forceinline__ float array_max(const float (&v)[N]) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) m = fmaxf(m, fabsf(v[i]));
    return m;
}
// The exponent an input may carry so that the OUTPUT scale 2^(es_in + ex + ew) stays a finite float with room for the
// accumulation (bias 2^es_out and sums of products): es_out <= 100.
__device__ __forceinline__ int cap_exp(int ex, int es_in, int ew) {
    const int room = 100 - es_in - ew;
    return ex < room ? ex : room;
}

// two fp32 values times 2^ex (r = pow2i(ex)) -> their (hi, lo) fp16 parts, packed pairwise into dword p of the two operands.
// NOP: this is the last pair of an operand that the very next MFMA may read -- a VGPR written inside inline asm needs two
// wait states before an MFMA reads it, which hipcc does not pad (field_common.h, relu1); once per k16-step.
// the same in two halves, for two different MFMA gaps: v_fma_mix* issues at half rate (tools/probe/issue_probe.hip: 9.3 cycles
// each, a gap hides ~24), so four of them in one gap overrun it
__device__ __forceinline__ void split_pair_hi(Split2& o, int p, float x0, float x1, float r) {
    uint32_t h;
    asm volatile(
        "v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
        : "=&v"(h)
        : "v"(x0), "v"(x1), "v"(r));
    o.h[p] = h;
}
template <bool NOP>
__device__ __forceinline__ void split_pair_lo(Split2& o, int p, float x0, float x1, float r) {
    uint32_t l;
    const uint32_t h = o.h[p];
    if (NOP)
        asm volatile(
            "v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "s_nop 1"
            : "=&v"(l)
            : "v"(x0), "v"(x1), "v"(r), "v"(h));
    else
        asm volatile(
            "v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(l)
            : "v"(x0), "v"(x1), "v"(r), "v"(h));
    o.l[p] = l;
}
template <bool NOP>
__device__ __forceinline__ void split_pair_h(Split2& o, int p, float x0, float x1, float r) {
    uint32_t h, l;
    if (NOP)
        asm volatile(
            "v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "s_nop 1"
            : "=&v"(h), "=&v"(l)
            : "v"(x0), "v"(x1), "v"(r));
    else
        asm volatile(
            "v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l)
            : "v"(x0), "v"(x1), "v"(r));
    o.h[p] = h;
    o.l[p] = l;
}

// ---- B-operand sources.  Values 2p, 2p+1 of k16-step q (source tile T0 + q/2, registers 8(q%2)..+7) are transformed, scaled
// by r = 2^ex and split in three STAGES that mma_run_h3 places in three different MFMA gaps (an MFMA hides about six other
// issues; a whole pair in one gap overruns it):
//     A  fetch the two values (accumulator reads, LDS reads) and touch the ReLU-mask word (capture forward, apply backward)
//     B  activation and the running maximum of the consumed values (field_h3.h header: next operand's exponent)
//     C  the four conversions into dword p of the (hi, lo) operand
// Mask words are walked exactly as in field_x6.h / field_common.h, so the fp16 kernels exchange ReLU masks with every other
// forward / backward kernel. ---------------------------------------------------------------------------------------------
struct PairRegs {
    float x0, x1;
};
// One value of a source tile for the vector ALU.  H3_ACC_READ_ASM (the Wd = 256 forward objects, whose MFMAs are asm statements on
// "+a" tiles): the tile stays in its AGPRs and the value is read HERE -- an explicit v_accvgpr_read_b32 inside the MFMA gap that
// hosts the pair.  Left to itself hipcc moves the whole source set (128 registers) to VGPRs in one block at every layer boundary,
// where the wave's matrix pipe idles: ~280 instructions between two runs, 7 % of the forward (seen in the disassembly).
__device__ __forceinline__ float acc_read(const float& x) {
#if defined(H3_ACC_READ_ASM)
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x));
    return v;
#else
    return x;
#endif
}
template <bool CAPTURE, int NX, int NWORDS, int T0 = 0>
struct ReluSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;                   // running max of the consumed values (accumulator units), this lane
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        s.x0 = acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]);
        s.x1 = acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]);
        if (CAPTURE) {
            mask_shift_in(bits[(8 * q + 2 * p) >> 5], s.x0);
            mask_shift_in(bits[(8 * q + 2 * p + 1) >> 5], s.x1);
        }
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const {
        s.x0 = relu1<false>(s.x0);
        s.x1 = relu1<false>(s.x1);
        max3_acc(m, s.x0, s.x1);
    }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
// ReluSplitH whose source tiles do NOT contain their layer's bias yet: the producing run started every tile from the constant 0
// (mma_run_h3_wide C0) instead of writing sixteen bias x 2^es values into its accumulators (a v_mul and a v_accvgpr_write each), and
// the bias is added HERE, where the value is in a VGPR anyway: x = fma(b, 2^es, acc) -- one instruction per value instead of two, and
// one rounding of the sum.  bp: the source layer's bias block (this lane half's +16 bytes included), bs = 2^es of the source set.
// The two bias values of a pair are one 8-byte LDS read, requested one pair ahead (`nb` carries it from call to call: the pairs of
// a run are visited in order, k-step by k-step), so its latency is a unit or more old when the fma needs it.
template <bool CAPTURE, int NX, int NWORDS>
struct ReluBiasSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;
    const char* bp;
    float bs;
    float2& nb;                 // bias pair of the NEXT stage_a call (primed by the caller with pair (0, 0))
    static __device__ __forceinline__ int pair_off(int q, int p) {             // byte offset of the pair's two floats in the block
        const int r0 = (q & 1) * 8 + 2 * p;
        return ((q >> 1) * 32 + 8 * (r0 >> 2) + (r0 & 3)) * 4;
    }
    static __device__ __forceinline__ float2 prime(const char* bp_) { return *(const float2*)(bp_ + pair_off(0, 0)); }
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        const float2 b = nb;
        const int qn = p == 3 ? q + 1 : q, pn = p == 3 ? 0 : p + 1;
        nb = *(const float2*)(bp + pair_off(qn, pn));                          // (past the last pair: the next block's first floats, unused)
        s.x0 = __builtin_fmaf(b.x, bs, acc_read(X[q >> 1][(q & 1) * 8 + 2 * p]));
        s.x1 = __builtin_fmaf(b.y, bs, acc_read(X[q >> 1][(q & 1) * 8 + 2 * p + 1]));
        if (CAPTURE) {
            mask_shift_in(bits[(8 * q + 2 * p) >> 5], s.x0);
            mask_shift_in(bits[(8 * q + 2 * p + 1) >> 5], s.x1);
        }
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const {
        s.x0 = relu1<false>(s.x0);
        s.x1 = relu1<false>(s.x1);
        max3_acc(m, s.x0, s.x1);
    }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
// IdentSplitH with the source layer's bias added on the way in (see ReluBiasSplitH): xyz_encoding_final read by the heads.
template <int NX>
struct IdentBiasSplitH {
    const f32x16 (&X)[NX];
    float r;
    float& m;
    const char* bp;
    float bs;
    float2& nb;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        const float2 b = nb;
        const int qn = p == 3 ? q + 1 : q, pn = p == 3 ? 0 : p + 1;
        nb = *(const float2*)(bp + ReluBiasSplitH<false, 1, 1>::pair_off(qn, pn));
        s.x0 = __builtin_fmaf(b.x, bs, acc_read(X[q >> 1][(q & 1) * 8 + 2 * p]));
        s.x1 = __builtin_fmaf(b.y, bs, acc_read(X[q >> 1][(q & 1) * 8 + 2 * p + 1]));
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const { absmax3_acc(m, s.x0, s.x1); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
template <int NX, int NWORDS, int T0>
struct MaskedSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        s.x0 = mask_pick<false>(bits[(8 * q + 2 * p) >> 5], 8 * q + 2 * p, acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]));
        s.x1 = mask_pick<false>(bits[(8 * q + 2 * p + 1) >> 5], 8 * q + 2 * p + 1, acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]));
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const {
        absmax3_acc(m, s.x0, s.x1);
    }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
// MaskedSplitH whose values get a per-lane addend first: x = acc + add[e] * sa (element e = 8 q + 2 p of the k-step; sa brings the
// addend to the accumulators' scale) -- the factored head's d loss / d g joining the colour head's transposed product in front of
// dir_encoding^T (field_bwd_h3.hip FH).
template <int NX, int NWORDS, int T0, int NADD>
struct MaskedAddSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;
    const float (&add)[NADD];
    float sa;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        const float x0 = __builtin_fmaf(add[8 * q + 2 * p], sa, acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]));
        const float x1 = __builtin_fmaf(add[8 * q + 2 * p + 1], sa, acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]));
        s.x0 = mask_pick<false>(bits[(8 * q + 2 * p) >> 5], 8 * q + 2 * p, x0);
        s.x1 = mask_pick<false>(bits[(8 * q + 2 * p + 1) >> 5], 8 * q + 2 * p + 1, x1);
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const { absmax3_acc(m, s.x0, s.x1); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
template <int NX, int T0>
struct IdentSplitH {
    const f32x16 (&X)[NX];
    float r;
    float& m;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        s.x0 = acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]);
        s.x1 = acc_read(X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]);
    }
    __device__ __forceinline__ void stage_b(PairRegs& s) const { absmax3_acc(m, s.x0, s.x1); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
template <int N>
struct ArraySplitH {            // per-lane values v[8q + i] (embedding slots); their maximum is known to the caller
    const float (&v)[N];
    float r;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const { s.x0 = v[8 * q + 2 * p]; s.x1 = v[8 * q + 2 * p + 1]; }
    __device__ __forceinline__ void stage_b(PairRegs&) const {}
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};
// per-lane values parked in LDS between their uses (the xyz embedding: layer 1 and the skip at layer 5): slot s of this lane
// at base[s * 64] (base already carries wave and lane: lane-consecutive dwords, conflict-free)
struct LdsSplitH {
    const float* base;
    float r;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        s.x0 = base[(8 * q + 2 * p) * 64];
        s.x1 = base[(8 * q + 2 * p + 1) * 64];
    }
    __device__ __forceinline__ void stage_b(PairRegs&) const {}
    template <bool NOP>
    __device__ __forceinline__ void stage_c(Split2& o, int p, const PairRegs& s) const { split_pair_h<NOP>(o, p, s.x0, s.x1, r); }
    __device__ __forceinline__ void stage_c1(Split2& o, int p, const PairRegs& s) const { split_pair_hi(o, p, s.x0, s.x1, r); }
    template <bool NOP>
    __device__ __forceinline__ void stage_c2(Split2& o, int p, const PairRegs& s) const { split_pair_lo<NOP>(o, p, s.x0, s.x1, r); }
};

// C operand of a tile's first MFMA: bias rows times 2^es_out (the lane's output scale)
struct BiasInitScaled {
    const char* p;              // bias block of the layer + 16*h bytes
    float s;
    __device__ __forceinline__ f32x16 operator()(int t) const {
        f32x16 c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4*)(p + (t * 32 + 8 * q) * 4);
            c[4 * q + 0] = b[0] * s; c[4 * q + 1] = b[1] * s; c[4 * q + 2] = b[2] * s; c[4 * q + 3] = b[3] * s;
        }
        return c;
    }
};

// Weight-stream ring filled THROUGH REGISTERS: each wave moves NEFES_SLAB_PIECES contiguous 1 KiB pieces of every slab with
// plain global_load_dwordx4 (into `stage`) and ds_write_b128, instead of global_load_lds_dwordx4 (WeightRing).  Issuing an
// LDS-DMA piece stalls the issuing wave for 60-185 cycles among MFMAs and ds_reads (MI355X_MICROARCH.md); in the fp16 kernels,
// whose units carry three MFMAs for 2 KiB of weights, that was a quarter of the kernel time (tools/ablate_h3.sh: 215 -> 164 ms
// without the transfers).  A register-staged piece costs two ordinary issues, the compiler counts the loads itself (no
// "no compiler load while a DMA is in flight" rule, field_common.h), and two slots suffice:
//     while slab i is consumed from slot i%2, piece q of slab i+1 (loaded during slab i-1, one slab time of latency cover) is
//     written to slot (i+1)%2 and the piece q of slab i+2 is requested into the same four registers;
//     acquire() = own writes done (lgkmcnt) + workgroup barrier: slab i+1 is complete and nobody reads slab i any more.
// Same interface as WeightRing towards mma_run / mma_run_h3.
struct StagedRing {
    const char* src;        // stream base (wave-uniform)
    uint32_t n_slabs;       // slabs in the stream (wraps)
    uint32_t g_next;        // slab the next loads fetch
    uint32_t c_slot;        // slot being consumed (0/1)
    uint32_t cur_off;       // its LDS offset
    uint32_t my_off;        // wave * PIECES KiB + lane * 16: this lane's share of every slab
    char* my_lds;           // ring base + my_off
    f32x4 pf;               // first fragment group of the slab being consumed
    f32x4 stage[NEFES_SLAB_PIECES];

    __device__ __forceinline__ void load_piece(int q) {
        stage[q] = *(const f32x4*)(src + (size_t)g_next * NEFES_SLAB_BYTES + my_off + q * 1024);
    }
    __device__ __forceinline__ void init(const char* stream, uint32_t nslabs, char* ring_base, int wave, int lane) {
        src = stream; n_slabs = nslabs;
        my_off = (uint32_t)(wave * NEFES_SLAB_PIECES * 1024 + lane * 16);
        my_lds = ring_base + my_off;
        g_next = 0; c_slot = 0; cur_off = 0;
#pragma unroll
        for (int q = 0; q < NEFES_SLAB_PIECES; ++q) load_piece(q);
#pragma unroll
        for (int q = 0; q < NEFES_SLAB_PIECES; ++q) *(f32x4*)(my_lds + q * 1024) = stage[q];      // slab 0 -> slot 0
        g_next = n_slabs > 1 ? 1 : 0;
#pragma unroll
        for (int q = 0; q < NEFES_SLAB_PIECES; ++q) load_piece(q);                                 // slab 1 in flight
        g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
    }
    // the two halves of issue_piece for callers that place them in different MFMA gaps (mma_run_h3_wide)
    __device__ __forceinline__ void store_piece(int q) {
        *(f32x4*)(my_lds + (c_slot ^ 1u) * NEFES_SLAB_BYTES + q * 1024) = stage[q];
    }
    __device__ __forceinline__ void fetch_piece(int q) {
        load_piece(q);
        if (q == NEFES_SLAB_PIECES - 1) g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
    }
    // piece q of the next slab: registers -> the idle slot; then request the same piece of the slab after it
    __device__ __forceinline__ void issue_piece(int q) {
        *(f32x4*)(my_lds + (c_slot ^ 1u) * NEFES_SLAB_BYTES + q * 1024) = stage[q];
        load_piece(q);
        if (q == NEFES_SLAB_PIECES - 1) g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
    }
    __device__ __forceinline__ uint32_t acquire() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        c_slot ^= 1u;
        return c_slot * NEFES_SLAB_BYTES;
    }
    __device__ __forceinline__ void prime(const char* ring_lane) {      // after init(): publish slab 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur_off = 0;
        pf = *(const f32x4*)(ring_lane);
    }
    __device__ __forceinline__ void drain() {}
};

// acc[T0 .. T0+NT) = W-block * src over KS16 steps of 16 k-values; init(t) is the C operand of each tile's first MFMA
// (FIRST = false: accumulate onto acc).  Stream order: for k16-step q, for tile t: [A_hi | A_lo] (2 KiB unit);
// floor(slab KiB / 2) units per slab; a segment starts on a slab boundary.
//
// One unit = three dependent MFMAs (small terms first: Al Bh, Ah Bl, Ah Bh) and three issue gaps.  The matrix pipe takes an MFMA
// every 32 cycles and an MFMA occupies the issue port for 8, so each gap hides about six other instructions -- IF they are there:
// the kernel's side work (operand split, weight-ring moves, A-operand reads, bias tiles) is placed gap by gap in source order and
// pinned with sched_barrier(0), instead of leaving a ten-instruction block per pair to the scheduler (measured: a single extra
// v_max3 per pair moved the kernel by 8-12 %: the old placement overran its gaps).  Per unit:
//     M1   gap 1: read next unit's A_hi            + stage A of a pair hosted by this tile
//     M2   gap 2: read next unit's A_lo (into al)  + stage B of that pair      + bias tile of the next output tile (first k-step)
//     M3   gap 3: stage C of that pair  -- or, on the units that host none, one weight-ring piece (write + load)
// The operand of k16-step q+1 is produced during step q: pair pp by tile pp * STRIDE (every second tile of a >= 8-tile segment), so
// its last conversion is at least one whole unit ahead of its first consumer and needs no wait-state padding there.
// The MFMA of the wide runs as a volatile asm statement: every other instruction of mma_run_h3_wide's unit is volatile asm or
// pinned between volatile statements, but an MFMA builtin is a pure value to the compiler, which linearises the block's DAG
// with the three MFMAs of a unit back to back and the side work in clumps -- whatever the source order and sched_barrier say
// (seen in the disassembly).  As asm the MFMAs ARE the skeleton of the instruction stream.  Accumulators live in AGPRs ("+a").
// Hazards the compiler no longer pads for (it does not look inside asm): an asm-written B operand needs two instructions before
// the MFMA that reads it (stage C2 of a pair is two MFMAs ahead of its consumer; the run's first operand ends in s_nop 1); a VALU
// read of a tile follows the tile's last MFMA by a whole layer, a VALU write of a bias tile precedes its first MFMA by a unit;
// operands returned by LDS reads are waited for by the compiler's own s_waitcnt (it tracks asm operands).
__device__ __forceinline__ void mfma_h3_asm(f32x16& c, const f32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// The same MFMA on a tile that lives in VGPRs: the backward's wide runs keep 16 tiles (source + destination set) in the 256 AGPRs
// and the few extra output tiles of a run (the d embedding tiles riding on layer 5) in vector registers.
__device__ __forceinline__ void mfma_h3_asm_v(f32x16& c, const f32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// First MFMA of a tile whose C operand is zero (the backward's transposed products): the inline constant 0 as srcC, the tile is an
// output only -- no sixteen v_accvgpr_write per tile and layer to zero it (a third of a VALU instruction per MFMA in the backward).
__device__ __forceinline__ void mfma_h3_asm_c0(f32x16& c, const f32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_h3_asm_v_c0(f32x16& c, const f32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}
template <class T> struct h3_is_zero_init { static constexpr bool value = false; };
template <> struct h3_is_zero_init<ZeroInit> { static constexpr bool value = true; };
// VT: tiles acc[0 .. VT) live in VGPRs ("+v" MFMAs), the rest in AGPRs.
template <int NT, int KS16, int T0, bool FIRST = true, int VT = 0, class SrcFn, class InitFn, int NACC, class Ring>
__device__ __forceinline__ void mma_run_h3_wide(Ring& ring, const char* ring_lane, const SrcFn& src, const InitFn& init,
                                                f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    static_assert(T0 + NT - 2 >= VT, "the run's last two tiles must be AGPR tiles (end-of-run wait states)");
    static_assert(NT >= 8 && NT % 2 == 0, "the gap schedule below pairs tiles: an even tile hosts a pair, the odd one a ring piece");
    constexpr int UPS = (NEFES_SLAB_FRAGS / 4) / 2;       // units per slab
    constexpr int NU = KS16 * NT;
    constexpr int NSLAB = (NU + UPS - 1) / UPS;
    // No register is copied inside the unit loop: a copy lands right behind an MFMA that may still be READING the copy's
    // destination as an operand (the compiler pads that write-after-read hazard for its own MFMAs, not for asm ones -- seen as
    // garbage results).  The operand of step q lives in B0 / B1 by the parity of q, unit u's weight groups in (ha, la)[u % 3];
    // everything is selected at compile time (the loops are fully unrolled).
    Split2 B0, B1;
    auto BQ = [&](int k) -> Split2& { return (k & 1) ? B1 : B0; };
    // (`s_nop 13` here and `s_nop 12` in the end-of-run statement occur nowhere else in the library: tests/test_pack_stream.py
    // finds the asm-scheduled stretches of the disassembly by them and checks that no accumulator tile is moved inside one)
    asm volatile("s_nop 13" ::: "memory");
#ifdef H3_WIDE_ENTRY_FENCE
    // The run's first operand is read out of the source tiles' AGPRs by asm statements, which hipcc schedules freely among its OWN
    // MFMAs: when the previous segment ran on compiler-placed MFMAs (narrow runs, the fp32 sigma step) it hoisted these reads to
    // right behind the MFMA that writes the register -- inside that MFMA's 16 passes, where the read returns the old value (seen in
    // the disassembly of the backward; wrong gradients).  Nothing crosses this point, and the youngest result is 18 wait states old.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    {
        PairRegs s0;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            src.stage_a(s0, 0, pp);
            src.stage_b(s0);
            if (pp == 3) src.template stage_c<true>(B0, pp, s0);
            else src.template stage_c<false>(B0, pp, s0);
        }
    }
    // FIRST: the C operand of a tile's first MFMA (bias x 2^es, or zero) is written straight into the tile's own registers,
    // one unit ahead of its first use -- the output tiles are dead until then -- rather than into a 16-register staging tile
    constexpr bool C0 = FIRST && h3_is_zero_init<InitFn>::value
        ;                                                  // zero C operand: the first MFMA of every tile takes the constant
    if (FIRST && !C0) acc[T0] = init(0);
    // A operands are requested TWO units (six MFMAs) ahead of their use; (h0,l0), (h1,l1) = this and the next unit's, the reads
    // in flight are the unit's after that.  ring.pf carries the first unit's hi group across segments.
    const char* p = ring_lane + ring.cur_off;
    f32x4 ha0 = ring.pf, la0 = *(const f32x4*)(p + 1024);
    f32x4 ha1 = *(const f32x4*)(p + 2048), la1 = *(const f32x4*)(p + 3072);
    f32x4 ha2, la2;
    auto HA = [&](int k) -> f32x4& { return k % 3 == 0 ? ha0 : (k % 3 == 1 ? ha1 : ha2); };
    auto LA = [&](int k) -> f32x4& { return k % 3 == 0 ? la0 : (k % 3 == 1 ? la1 : la2); };
    PairRegs ps[4];                                        // pairs in flight between their stages
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
        // The slab after this one is acquired in unit acq_u (two units before the end: the request for unit u+2 then crosses into
        // it).  Ring piece qq is written to LDS and re-requested in the odd unit piece_unit(qq); what is left goes in front of the acquire.
        const int acq_u = nu >= 2 ? nu - 2 : 0;
        auto piece_unit = [&](int qq) { const int m_ = ((2 * qq + 1) * nu) / UPS; return m_ < acq_u ? m_ : acq_u; };
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                const bool make = q + 1 < KS16;            // this step produces the next step's operand
                const int pp = t / 2;
                const bool host = make && (t % 2 == 0) && pp < 4;    // pair t/2: stages A, B, C1 here, C2 in the next (odd) unit
                const bool tail = make && (t % 2 == 1) && pp < 4;    // (a k-step has four pairs; a ten-tile run's last two tiles host none)
                const char* pn = p + (2 * uu + 4) * 1024;  // unit u + 2's hi group (lo: + 1 KiB)
                Split2& B = BQ(q);
                Split2& Bn = BQ(q + 1);
                f32x4 &h0 = HA(u), &l0 = LA(u), &h2 = HA(u + 2), &l2 = LA(u + 2);
                __builtin_amdgcn_sched_barrier(0);
                // the tile's C operand was written by the vector ALU (bias tile): materialise it in its AGPRs HERE, then two wait states
                const bool vt = T0 + t < VT;               // (folds: the loops are fully unrolled)
                if (C0 && q == 0) {
                    if (vt) mfma_h3_asm_v_c0(acc[T0 + t], l0, B.h);
                    else mfma_h3_asm_c0(acc[T0 + t], l0, B.h);
                } else {
                    if (FIRST && q == 0) {
                        if (vt) asm volatile("s_nop 1" : "+v"(acc[T0 + t]));
                        else asm volatile("s_nop 1" : "+a"(acc[T0 + t]));
                    }
                    if (vt) mfma_h3_asm_v(acc[T0 + t], l0, B.h);
                    else mfma_h3_asm(acc[T0 + t], l0, B.h);                                         // M1 (small terms first)
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- gap 1 ----
                if (host) src.stage_a(ps[pp], q + 1, pp);
                if (tail) src.template stage_c2<false>(Bn, pp, ps[pp]);
                if (!host && uu != acq_u) h2 = *(const f32x4*)(pn);
                __builtin_amdgcn_sched_barrier(0);
                // An asm's inputs are dead to the compiler once it has issued, but the MFMA is still reading them: without these
                // empty uses the registers are handed to the gap's instructions and overwritten under the MFMA (seen in the
                // disassembly as conversions writing the previous MFMA's operand registers; the results were garbage).
                asm volatile("" ::"v"(l0), "v"(B.h));
                if (vt) mfma_h3_asm_v(acc[T0 + t], h0, B.l);
                else mfma_h3_asm(acc[T0 + t], h0, B.l);                                             // M2
                __builtin_amdgcn_sched_barrier(0);
                // ---- gap 2 ----
                if (host) src.stage_b(ps[pp]);
                if (host && uu != acq_u) h2 = *(const f32x4*)(pn);
                if (!host) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if (piece_unit(qq) == uu) ring.store_piece(qq);
                }
                if (FIRST && !C0 && q == 0 && t + 1 < NT) {
                    acc[T0 + t + 1] = init(t + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::"v"(B.l));
                if (vt) mfma_h3_asm_v(acc[T0 + t], h0, B.h);
                else mfma_h3_asm(acc[T0 + t], h0, B.h);                                             // M3
                __builtin_amdgcn_sched_barrier(0);
                // ---- gap 3 ----
                if (host) src.stage_c1(Bn, pp, ps[pp]);
                if (!host) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if (piece_unit(qq) == uu) ring.fetch_piece(qq);
                }
                if (uu == acq_u) {                         // next slab: everyone's pieces written, nobody requests from this one any more
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if (piece_unit(qq) == uu && host) { ring.store_piece(qq); ring.fetch_piece(qq); }
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off - (size_t)nu * 2048;      // so that unit index uu + 2 >= nu addresses the new slab
                    pn = p + (2 * uu + 4) * 1024;
                    h2 = *(const f32x4*)(pn);
                }
                l2 = *(const f32x4*)(pn + 1024);
                if (uu + 1 == nu) p = ring_lane + ring.cur_off;            // plain addressing again from the new slab's unit 0
                asm volatile("" ::"v"(h0), "v"(B.h));                      // M3's operands stay untouched through its gap
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    ring.pf = HA(NU);
    // The layer's functor reads the tiles with the vector ALU: 18 wait states between an MFMA and such a read of its result are the
    // program's to provide (the compiler pads them for its own MFMAs only).  Tying the two youngest tiles to the statement keeps their
    // reads behind it; the older tiles' last MFMAs are at least six MFMAs back.
    // The same statement holds the LAST MFMA's A and B operand registers: an issued asm MFMA is still reading them, and once the
    // unit's empty use is behind it hipcc hands them out again -- the very next instruction overwrote the B operand (seen in the
    // disassembly of the backward: v_mfma ... v[14:17] followed by v_mov_b32 v14; the last tile's last k-step came out wrong).
    asm volatile("s_nop 12\n\ts_nop 4"
                 : "+a"(acc[T0 + NT - 1]), "+a"(acc[T0 + NT - 2])
                 : "v"(HA(NU - 1)), "v"(BQ(KS16 - 1).h), "v"(BQ(KS16 - 1).l));
    // The run's VGPR tiles keep their registers through those wait states even when nobody reads them afterwards (the second d
    // embedding tile of the instances with an external / hash-grid encoding is such a tile): a dead tile's registers were handed to
    // the next ring loads ONE wait state behind the asm MFMA still writing them (tools/hazard_lint.py rule B2, write after write).
#pragma unroll
    for (int t = 0; t < VT; ++t)
        if (t >= T0 && t < T0 + NT) asm volatile("" : "+v"(acc[t]));
}

// Narrow segments (fewer than eight tiles: the half-width layers and the heads, a few per cent of the MFMAs): the operand of the
// next step is produced in one block behind a tile's third MFMA and left to the scheduler's interleaving hints.  (The gap-by-gap
// form above, applied to them, made hipcc sink the whole segment's MFMAs behind its operand production and spill every A operand.)
template <int NT, int KS16, int T0, bool FIRST = true, class SrcFn, class InitFn, int NACC, class Ring>
__device__ __forceinline__ void mma_run_h3_small(Ring& ring, const char* ring_lane, const SrcFn& src, const InitFn& init,
                                           f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    constexpr int UPS = (NEFES_SLAB_FRAGS / 4) / 2;       // units per slab
    constexpr int NU = KS16 * NT;
    constexpr int NSLAB = (NU + UPS - 1) / UPS;
    Split2 B, Bn;
    {
        PairRegs s0;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            src.stage_a(s0, 0, pp);
            src.stage_b(s0);
            if (pp == 3) src.template stage_c<true>(B, pp, s0);
            else src.template stage_c<false>(B, pp, s0);
        }
    }
    Bn = B;
    // FIRST: the C operand of a tile's first MFMA (bias x 2^es, or zero) is written straight into the tile's own registers,
    // one unit ahead of its first use -- the output tiles are dead until then -- rather than into a 16-register staging tile
    // (a 16-register staging tile per run was what this kernel family could not afford before the source tiles stayed in AGPRs)
    if (FIRST) acc[T0] = init(0);
    const char* p = ring_lane + ring.cur_off;
    f32x4 ah = ring.pf, al = *(const f32x4*)(p + 1024);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                // A operands of the next unit (of this slab or of the one acquired here): the hi group is requested now, the lo
                // group behind this unit's first MFMA -- the only reader of `al` -- straight into `al` (four registers less)
                f32x4 nh;
                const bool last = !(uu + 1 < nu);
                if (!last) {
                    nh = *(const f32x4*)(p + (2 * uu + 2) * 1024);
                } else {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES >= uu) ring.issue_piece(qq);   // everything still owed to this slab
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off;
                    nh = *(const f32x4*)(p);
                }
                if (t == 0 && q > 0) B = Bn;
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 Ah = as_f16x8(ah), Al = as_f16x8(al);
                const f16x8 Bh = as_f16x8(B.h), Bl = as_f16x8(B.l);
                f32x16 c = acc[T0 + t];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, c, 0, 0, 0);          // small terms first
                al = *(const f32x4*)(p + (last ? 1 : 2 * uu + 3) * 1024);
                if (FIRST && q == 0 && t + 1 < NT) acc[T0 + t + 1] = init(t + 1);
                if (uu + 1 < nu) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES == uu) {
                            __builtin_amdgcn_sched_barrier(0);
                            ring.issue_piece(qq);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, c, 0, 0, 0);
                acc[T0 + t] = c;
                if (q + 1 < KS16) {
                    constexpr int STRIDE = NT >= 8 ? 2 : 1;                   // tiles between two pairs
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        if (NT >= 4 ? (t == pp * STRIDE + STRIDE - 1) : (t == (pp * NT) / 4)) {
                            PairRegs s1;
                            src.stage_a(s1, q + 1, pp);
                            src.stage_b(s1);
                            if (pp == 3) src.template stage_c<true>(Bn, pp, s1);
                            else src.template stage_c<false>(Bn, pp, s1);
#pragma unroll
                            for (int i = 0; i < 2; ++i) {                     // interleave: one MFMA, then up to four VALU
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            }
                        }
                }
                ah = nh;
            }
        }
    }
    ring.pf = ah;
    mfma_results_fence_tiles<12, NT, T0>(acc);
}

template <int NT, int KS16, int T0, bool FIRST = true, int VT = 0, bool ALLOW_WIDE = true, class SrcFn, class InitFn, int NACC, class Ring>
__device__ __forceinline__ void mma_run_h3(Ring& ring, const char* ring_lane, const SrcFn& src, const InitFn& init,
                                           f32x16 (&acc)[NACC]) {
    if constexpr (ALLOW_WIDE && NT >= NEFES_H3_WIDE_MIN && NT % 2 == 0 && NT <= NEFES_H3_WIDE_MAX) mma_run_h3_wide<NT, KS16, T0, FIRST, VT>(ring, ring_lane, src, init, acc);
    else mma_run_h3_small<NT, KS16, T0, FIRST>(ring, ring_lane, src, init, acc);
}
