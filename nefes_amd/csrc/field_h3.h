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

// ---- B-operand sources: pair(o, q, p) = values 2p, 2p+1 of k16-step q (source tile T0 + q/2, registers 8(q%2)..+7),
// transformed, scaled by r = 2^ex and split.  Mask words are walked exactly as in field_x6.h / field_common.h, so the fp16
// kernels exchange ReLU masks with every other forward / backward kernel. ---------------------------------------------------
template <bool CAPTURE, int NX, int NWORDS, int T0 = 0>
struct ReluSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;                   // running max of the consumed values (accumulator units), this lane
    template <bool NOP>
    __device__ __forceinline__ void pair(Split2& o, int q, int p) const {
        const float v0 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p], v1 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1];
        if (CAPTURE) {
            mask_shift_in(bits[(8 * q + 2 * p) >> 5], v0);
            mask_shift_in(bits[(8 * q + 2 * p + 1) >> 5], v1);
        }
        const float x0 = relu1<false>(v0), x1 = relu1<false>(v1);
        max3_acc(m, x0, x1);
        split_pair_h<NOP>(o, p, x0, x1, r);
    }
};
template <int NX, int NWORDS, int T0>
struct MaskedSplitH {
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    float r;
    float& m;
    template <bool NOP>
    __device__ __forceinline__ void pair(Split2& o, int q, int p) const {
        const float x0 = mask_shift_out<false>(bits[(8 * q + 2 * p) >> 5], X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]);
        const float x1 = mask_shift_out<false>(bits[(8 * q + 2 * p + 1) >> 5], X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]);
        absmax3_acc(m, x0, x1);
        split_pair_h<NOP>(o, p, x0, x1, r);
    }
};
template <int NX, int T0>
struct IdentSplitH {
    const f32x16 (&X)[NX];
    float r;
    float& m;
    template <bool NOP>
    __device__ __forceinline__ void pair(Split2& o, int q, int p) const {
        const float x0 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p], x1 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1];
        absmax3_acc(m, x0, x1);
        split_pair_h<NOP>(o, p, x0, x1, r);
    }
};
template <int N>
struct ArraySplitH {            // per-lane values v[8q + i] (embedding slots)
    const float (&v)[N];
    float r;
    template <bool NOP>
    __device__ __forceinline__ void pair(Split2& o, int q, int p) const { split_pair_h<NOP>(o, p, v[8 * q + 2 * p], v[8 * q + 2 * p + 1], r); }
};

// per-lane values parked in LDS between their uses (the xyz embedding: layer 1 and the skip at layer 5): slot s of this lane
// at base[s * 64] (base already carries wave and lane: lane-consecutive dwords, conflict-free)
struct LdsSplitH {
    const float* base;
    float r;
    template <bool NOP>
    __device__ __forceinline__ void pair(Split2& o, int q, int p) const {
        split_pair_h<NOP>(o, p, base[(8 * q + 2 * p) * 64], base[(8 * q + 2 * p + 1) * 64], r);
    }
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
#ifdef H3_DBG_NOBIASMUL
            c[4 * q + 0] = b[0]; c[4 * q + 1] = b[1]; c[4 * q + 2] = b[2]; c[4 * q + 3] = b[3];
#else
            c[4 * q + 0] = b[0] * s; c[4 * q + 1] = b[1] * s; c[4 * q + 2] = b[2] * s; c[4 * q + 3] = b[3] * s;
#endif
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
// (FIRST = false: accumulate onto acc).
template <int NT, int KS16, int T0, bool FIRST = true, class SrcFn, class InitFn, int NACC, class Ring>
__device__ __forceinline__ void mma_run_h3(Ring& ring, const char* ring_lane, const SrcFn& src, const InitFn& init,
                                           f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    constexpr int UPS = (NEFES_SLAB_FRAGS / 4) / 2;       // units per slab
    constexpr int NU = KS16 * NT;
    constexpr int NSLAB = (NU + UPS - 1) / UPS;
    Split2 B, Bn;
    // H3_ABL_* macros: timing ablations only (tools/ablate_h3.sh builds side libraries with them; results are garbage)
    src.template pair<false>(B, 0, 0);
    src.template pair<false>(B, 0, 1);
    src.template pair<false>(B, 0, 2);
    src.template pair<true>(B, 0, 3);
    Bn = B;
    // FIRST: the C operand of a tile's first MFMA (bias x 2^es, or zero) is written straight into the tile's own registers,
    // one unit ahead of its first use -- the output tiles are dead until then -- rather than into a 16-register staging tile
    // (this kernel family sits a handful of registers below its 512: DESIGN.md)
#ifdef H3_ABL_NOBIAS
    if (FIRST) acc[T0] = ZeroInit{}(0);
#else
    if (FIRST) acc[T0] = init(0);
#endif
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
#ifdef H3_ABL_NOAREAD
                    nh = ah;
#else
                    nh = *(const f32x4*)(p + (2 * uu + 2) * 1024);
#endif
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
#ifndef H3_ABL_NOAREAD
                al = *(const f32x4*)(p + (last ? 1 : 2 * uu + 3) * 1024);
#endif
#ifdef H3_ABL_NOBIAS
                if (FIRST && q == 0 && t + 1 < NT) acc[T0 + t + 1] = ZeroInit{}(0);
#else
                if (FIRST && q == 0 && t + 1 < NT) acc[T0 + t + 1] = init(t + 1);
#endif
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
#ifndef H3_ABL_NOSPLIT
                            if (pp == 3) src.template pair<true>(Bn, q + 1, pp);
                            else src.template pair<false>(Bn, q + 1, pp);
#endif
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
}
