// bf16x6 split products on v_mfma_f32_32x32x16_bf16 for the 256-wide hidden layers (shared by field_fwd_x6.hip and
// field_bwd_x6.hip; include after field_common.h, with NEFES_SLAB_KIB defined by the translation unit).
//     x = xh + xm + xl, w = wh + wm + wl exactly (three bf16 each: 24 = 3 x 8 mantissa bits, truncation split)
//     w x ~= wh xh + wh xm + wm xh + wh xl + wl xh + wm xm        (dropped terms: relative size 2^-24)
// accumulated in fp32 by the matrix core: fp32-level accuracy (tools/bf16x6_accuracy.py) at 6 x 32 = 192 cycles per 16
// k-values and tile instead of 8 x 64 = 512.  The accumulator -> B-operand chaining carries over: a 32-row D tile is two
// K=16 steps (registers 8j..8j+7 of lane group g hold rows rho_g(8j+i)); the host packer emits the weight triples in that
// order (pack.cpp, x6 segments).  The bf16 MFMA holds the vector issue port for a quarter of its time, so the split
// (5.5 VALU per element) hides in the gaps (tools/probe/overlap_probe.hip).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
    u32x4 h, m, l;           // 8 bf16 each: element i in the low/high half of dword i/2
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float x0, float x1) {      // v_cvt_pk_bf16_f32: (RNE(x1) << 16) | RNE(x0)
    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
    uint32_t b;
    __builtin_memcpy(&b, &h, 4);
    return b;
}

// two fp32 values -> their (hi, mid, lo) bf16 parts packed pairwise into dword p of the three operand vectors.
// (A round-to-nearest split on v_cvt_pk_bf16_f32 + packed subtractions is 9 instead of 11 VALU per pair and measured
// 0.5 % slower per frame: the packed instructions take two passes.)
__device__ __forceinline__ void split_pair(Split3& o, int p, float x0, float x1) {
    {
        const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
        const float e0 = x0 - __uint_as_float(b0 & 0xffff0000u), e1 = x1 - __uint_as_float(b1 & 0xffff0000u);   // exact
        const uint32_t c0 = __float_as_uint(e0), c1 = __float_as_uint(e1);
        const float f0 = e0 - __uint_as_float(c0 & 0xffff0000u), f1 = e1 - __uint_as_float(c1 & 0xffff0000u);   // exact, <= 8 bits
        // v_perm_b32: bytes {src0 = element 1, src1 = element 0}; take the upper halves -> (hi16(x1) << 16) | hi16(x0)
        o.h[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        o.m[p] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
        o.l[p] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
    }
}

// Three-product mode: hi = RNE_bf16(x), mid = RNE_bf16(x - hi) (the residual is exact), so hi + mid is x rounded to ~16
// significant bits with a zero-mean error (a truncation split leaves a one-sided remainder that adds up coherently over
// the layers: measured 9e-5 instead of 2.6e-5 of the output scale).  v_cvt_pk_bf16_f32 converts a pair per instruction and the
// residuals are one packed subtraction: 6 VALU per pair.
__device__ __forceinline__ void split_pair16(Split3& o, int p, float x0, float x1) {
    const uint32_t hb = cvt_pk_bf16(x0, x1);
    const float e0 = x0 - __uint_as_float(hb << 16), e1 = x1 - __uint_as_float(hb & 0xffff0000u);    // exact
    o.h[p] = hb;
    o.m[p] = cvt_pk_bf16(e0, e1);
}
template <int NP>
__device__ __forceinline__ void split_pair_n(Split3& o, int p, float x0, float x1) {
    if constexpr (NP == 3) split_pair16(o, p, x0, x1);
    else split_pair(o, p, x0, x1);
}

// B-operand sources: get(q) = the 8 values of k16-step q (source tile T0 + q/2, registers 8(q%2) .. +7), transformed and
// split.  The mask-touching ones walk the activations in the same order as the fp32 functors (activation 8q + i <-> k-step
// 16T + r there), so forward and backward kernels of either kind exchange identical ReLU-mask words.
template <bool CAPTURE, int NX, int NWORDS, int T0 = 0>
struct ReluSplit {              // forward: relu(X) (+ mask capture), source tiles T0..
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    __device__ __forceinline__ void vals(int q, int p, float& x0, float& x1) const {   // values 2p, 2p+1 of k16-step q
        const float v0 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p], v1 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1];
        if (CAPTURE) {
            mask_shift_in(bits[(8 * q + 2 * p) >> 5], v0);
            mask_shift_in(bits[(8 * q + 2 * p + 1) >> 5], v1);
        }
        x0 = relu1<false>(v0); x1 = relu1<false>(v1);                          // one v_max each (fmaxf canonicalises first)
    }
    template <int NP = 6>
    __device__ __forceinline__ void pair(Split3& o, int q, int p) const { float x0, x1; vals(q, p, x0, x1); split_pair_n<NP>(o, p, x0, x1); }
};
template <int NX, int NWORDS, int T0>
struct MaskedSplit {            // backward: mask bit ? X : 0
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NWORDS];
    __device__ __forceinline__ void vals(int q, int p, float& x0, float& x1) const {
        x0 = mask_pick<false>(bits[(8 * q + 2 * p) >> 5], 8 * q + 2 * p, X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]);
        x1 = mask_pick<false>(bits[(8 * q + 2 * p + 1) >> 5], 8 * q + 2 * p + 1, X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1]);
    }
    template <int NP = 6>
    __device__ __forceinline__ void pair(Split3& o, int q, int p) const { float x0, x1; vals(q, p, x0, x1); split_pair_n<NP>(o, p, x0, x1); }
};
template <int NX, int T0>
struct IdentSplit {             // X as is
    const f32x16 (&X)[NX];
    __device__ __forceinline__ void vals(int q, int p, float& x0, float& x1) const {
        x0 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p]; x1 = X[T0 + (q >> 1)][(q & 1) * 8 + 2 * p + 1];
    }
    template <int NP = 6>
    __device__ __forceinline__ void pair(Split3& o, int q, int p) const { float x0, x1; vals(q, p, x0, x1); split_pair_n<NP>(o, p, x0, x1); }
};

// acc[T0 .. T0+NT) = W-block * src over KS16 steps of 16 k-values; init(t) (bias or zero) is the C operand of each tile's
// first MFMA.  Stream order: for k16-step q, for tile t: [A_hi | A_mid | A_lo] (3 KiB unit); floor(slab KiB / 3) units per slab.
// TRAIN instances of the backward kernel: store the consumed gradient values (see Storing in field_common.h)
template <class Inner>
struct StoringSplit {
    Inner in;
    float* p;
    template <int NP = 6>
    __device__ __forceinline__ void pair(Split3& o, int q, int pp) const {
        float x0, x1;
        in.vals(q, pp, x0, x1);
        const int s = 8 * q + 2 * pp;
        p[(s >> 4) * 4096 + nefes_rho(0, s & 15) * 16] = x0;      // layout.h nefes_train_off
        p[((s + 1) >> 4) * 4096 + nefes_rho(0, (s + 1) & 15) * 16] = x1;
        split_pair_n<NP>(o, pp, x0, x1);
    }
};
template <bool ON, class Inner>
__device__ __forceinline__ auto wrap_store_x6(const Inner& in, float* p) {
    if constexpr (ON) return StoringSplit<Inner>{in, p};
    else return in;
}

template <int N>
struct ArraySplit {             // per-lane values v[8q + i] (embedding slots) as they are
    const float (&v)[N];
    template <int NP = 6>
    __device__ __forceinline__ void pair(Split3& o, int q, int p) const { split_pair_n<NP>(o, p, v[8 * q + 2 * p], v[8 * q + 2 * p + 1]); }
};

// FIRST = false: accumulate onto acc (init unused).
// NP = 6: all six products (fp32-level accuracy).  NP = 3: hi*hi + hi*mid + mid*hi only (operands carried to 16 bits, error
// ~2^-16 of the product scale; the lo parts and their three MFMAs drop out, the split of the lo part is dead code).
template <int NT, int KS16, int T0, bool FIRST = true, int NP = 6, class SrcFn, class InitFn, int NACC, int SLOTS>
__device__ __forceinline__ void mma_run_x6(WeightRing<SLOTS>& ring, const char* ring_lane, const SrcFn& src,
                                           const InitFn& init, f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    constexpr int UPS = (NEFES_SLAB_FRAGS / 4) / 3;       // units per slab
    constexpr int NU = KS16 * NT;
    constexpr int NSLAB = (NU + UPS - 1) / UPS;
    // the operand of k16-step q+1 is produced during step q, one pair of values per tile (every second tile for NT >= 8),
    // so that pair's ~17 VALU instructions sit in the gaps of one unit's six MFMAs (3-4 per gap)
    Split3 B, Bn;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) src.template pair<NP>(B, 0, pp);
    Bn = B;
    f32x16 c0;                                             // bias tile of the next first-step unit, fetched one unit ahead
    if (FIRST) c0 = init(0);
    const char* p = ring_lane + ring.cur_off;
    f32x4 ah = ring.pf, am = *(const f32x4*)(p + 1024), al = *(const f32x4*)(p + 2048);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                // A operands of the next unit (of this slab, or of the slab acquired here: the stream is one sequence)
                f32x4 nh, nm, nl;
                if (uu + 1 < nu) {
                    nh = *(const f32x4*)(p + (3 * uu + 3) * 1024);
                    nm = *(const f32x4*)(p + (3 * uu + 4) * 1024);
                    nl = *(const f32x4*)(p + (3 * uu + 5) * 1024);
                } else {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES >= uu) ring.issue_piece(qq);   // everything still owed to this slab
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off;
                    nh = *(const f32x4*)(p);
                    nm = *(const f32x4*)(p + 1024);
                    nl = *(const f32x4*)(p + 2048);
                }
                if (t == 0 && q > 0) B = Bn;
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 Ah = as_bf16x8(ah), Am = as_bf16x8(am), Al = as_bf16x8(al);
                const bf16x8 Bh = as_bf16x8(B.h), Bm = as_bf16x8(B.m), Bl = as_bf16x8(B.l);
                f32x16 c;
                if (FIRST && q == 0) {
                    c = c0;
                    if (t + 1 < NT) c0 = init(t + 1);
                } else {
                    c = acc[T0 + t];
                }
                if constexpr (NP == 6) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, c, 0, 0, 0);      // small terms first
                else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, c, 0, 0, 0);
                if (uu + 1 < nu) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES == uu) {
                            __builtin_amdgcn_sched_barrier(0);
                            ring.issue_piece(qq);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                if constexpr (NP == 6) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, c, 0, 0, 0);
                acc[T0 + t] = c;
                if (q + 1 < KS16) {
                    constexpr int STRIDE = NT >= 8 ? 2 : 1;                   // tiles between two pairs
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        if (NT >= 4 ? (t == pp * STRIDE + STRIDE - 1) : (t == (pp * NT) / 4)) {
                            src.template pair<NP>(Bn, q + 1, pp);
                            // interleave: one MFMA, then up to four VALU instructions, five times
#pragma unroll
                            for (int i = 0; i < (NP == 6 ? 5 : 2); ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            }
                        }
                }
                ah = nh; am = nm; al = nl;
            }
        }
    }
    ring.pf = ah;
    mfma_results_fence<12>();
}

