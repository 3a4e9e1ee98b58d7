// EXPERIMENT (round 2): the sigma-only forward of the 8x256 network on v_mfma_f32_16x16x32_f16.
//
// Under this chip's power limit the 16x16x32 MFMA sustains 1.15 x the rate of the 32x32x16 one (tools/probe/clock_probe.hip).
// This kernel measures what that is worth inside a real field kernel before the FULL forward and the backward are re-laid out:
// same arithmetic as field_fwd_h3_kernel<SIGMA> (fp16 two-part split, three products, bound-based exponents), different tiling.
//
//   workgroup = 8 waves x 16 samples (128 samples per pass over the weight stream, as before); lane = (j = l % 16, g = l / 16):
//   sample j of the wave, lane group g.  A 16x16 C tile holds rows 16 t + 4 g + r (r = 0..3) of the sample: 2 x 16 tiles x 4 =
//   128 accumulator registers, so TWO waves fit a SIMD and hide each other's vector work (no hand-placed gap schedule here).
//   The operand of a 32-k step q is the eight registers of tiles 2q and 2q+1: logical k = 8 g + e  <->  input row
//   32 q + 16 (e / 4) + 4 g + e % 4, a column permutation the packer applies (h4_col below).
//   Weight unit = [A_hi | A_lo] of (32-k step, 16-row tile): 2 KiB, lane (i = l % 16, g) holds W[16 t + i][h4_col(q, g, 0..7)].
//   Price: every one of the 8 waves reads every A operand from LDS (twice the traffic of the 4-wave kernels).
//
// Self-contained on purpose: its own small blob (nefes_h4_sigma_pack), no entry in NefesBlobInfo, not reachable from render().
// tests/test_gpu_h4.py checks it against the float64 oracle and times it against the production kernel.
#define NEFES_SLAB_KIB 32
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "field_common.h"
#include "../../include/nefes_hip.h"

#include "field_x6.h"
#include "field_h3.h"

// Built twice (Makefile): H4_W = 256 with 8 waves per workgroup (the headline network; this object also holds layout (a) and the C
// entry points) and H4_W = 128 with 16 waves (the width every configuration file of the reference uses: 2 x 8 tiles x 4 = 64
// accumulator registers per wave, four waves per SIMD, 256 samples per pass over the weight stream).
#ifndef H4_W
#define H4_W 256
#endif
#ifndef H4_NW
#define H4_NW 8
#endif
namespace {

constexpr int kW = H4_W, kNW = H4_NW, kPieces = 32 / kNW, kNT = kW / 16, kKSH = kW / 32, kKSE = 2, kSegs = 10, kBiasBlocks = 9;
constexpr int kBiasFloats = 8 * kW + 16;                       // L1..L8, then the sigma head's tile (row 0 real)
constexpr int kTabOff = kBiasFloats;                            // (exp, bound) per segment, then max |b| per bias block
constexpr int kBlobFloats = ((kTabOff + 2 * kSegs + kBiasBlocks + 63) / 64) * 64;
enum { S_L1 = 0, S_L2, S_L3, S_L4, S_L5H, S_L5E, S_L6, S_L7, S_L8, S_SIG };

// feature of embedding slot u (0..15) of lane group g: pairs (frequency k, component a) n = 3 k + a are dealt round-robin to the
// groups (n = g + 4 i), slot 2i = sin, 2i+1 = cos; the two free slots of groups 2 and 3 carry x, y and z.  -1 = padding.
__host__ __device__ inline int h4_feature(int g, int u) {
    const int n = g + 4 * (u >> 1);
    if (n < 30) return 3 + 6 * (n / 3) + ((u & 1) ? 3 : 0) + n % 3;
    if (g == 2) return u & 1;             // x, y
    return (u & 1) ? -1 : 2;              // z, pad
}
__host__ __device__ inline int h4_col(int q, int g, int e) { return 32 * q + 16 * (e >> 2) + 4 * g + (e & 3); }

struct H4Args {
    const char* stream;
    const float* bias;
    uint32_t n_slabs;
    const float *rays_o, *rays_d, *z;
    float* raw_t;
    int N, S;
    long long M;
    int n_tiles;
    uint32_t s_magic, s_shift;
};

// weight ring through registers for kNW waves: each wave moves kPieces 1 KiB pieces of every 32 KiB slab (field_h3.h StagedRing)
struct Ring8 {
    const char* src;
    uint32_t n_slabs, g_next, c_slot, cur_off, my_off;
    char* my_lds;
    f32x4 pf;
    f32x4 stage[kPieces];
    __device__ __forceinline__ void load_piece(int q) { stage[q] = *(const f32x4*)(src + (size_t)g_next * 32768 + my_off + q * 1024); }
    __device__ __forceinline__ void init(const char* stream, uint32_t nslabs, char* ring_base, int wave, int lane) {
        src = stream; n_slabs = nslabs;
        my_off = (uint32_t)(wave * (kPieces * 1024) + lane * 16);
        my_lds = ring_base + my_off;
        g_next = 0; c_slot = 0; cur_off = 0;
#pragma unroll
        for (int q = 0; q < kPieces; ++q) load_piece(q);
#pragma unroll
        for (int q = 0; q < kPieces; ++q) *(f32x4*)(my_lds + q * 1024) = stage[q];
        g_next = n_slabs > 1 ? 1 : 0;
#pragma unroll
        for (int q = 0; q < kPieces; ++q) load_piece(q);
        g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
    }
    __device__ __forceinline__ void issue_piece(int q) {
        *(f32x4*)(my_lds + (c_slot ^ 1u) * 32768 + q * 1024) = stage[q];
        load_piece(q);
        if (q == kPieces - 1) g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
    }
    __device__ __forceinline__ uint32_t acquire() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        c_slot ^= 1u;
        return c_slot * 32768;
    }
    __device__ __forceinline__ void prime(const char* ring_lane) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur_off = 0;
        pf = *(const f32x4*)(ring_lane);
    }
};

// The split of one pair as NINE single instructions (layout (a) places them one per 16-cycle MFMA gap):
//   m0/m1: fetch x0 / x1     m2/m3: ReLU     m4: running maximum     m5/m6: hi halves     m7/m8: lo halves
struct MicroPair { float x0, x1; uint32_t h, l; };
__device__ __forceinline__ void mix_hi_lo(uint32_t& h, float x, float r) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x), "v"(r)); }
__device__ __forceinline__ void mix_hi_hi(uint32_t& h, float x, float r) { asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x), "v"(r)); }
__device__ __forceinline__ void mix_lo_lo(uint32_t& l, float x, float r, uint32_t h) {
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x), "v"(r), "v"(h));
}
__device__ __forceinline__ void mix_lo_hi(uint32_t& l, float x, float r, uint32_t h) {
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x), "v"(r), "v"(h));
}
// B-operand sources on 16-row tiles: pair p (0..3) of 32-k step q = registers 2 (p & 1), +1 of tile T0 + 2 q + (p >> 1)
template <int NX, int T0 = 0>
struct ReluSplit4 {
    const f32x4 (&X)[NX];
    float r;
    float& m;
    __device__ __forceinline__ void stage_a(PairRegs& s, int q, int p) const {
        s.x0 = X[T0 + 2 * q + (p >> 1)][2 * (p & 1)];
        s.x1 = X[T0 + 2 * q + (p >> 1)][2 * (p & 1) + 1];
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
    __device__ __forceinline__ void micro(int k, MicroPair& s, Split2& o, int q, int p) const {
        if (k == 0) s.x0 = X[T0 + 2 * q + (p >> 1)][2 * (p & 1)];
        if (k == 1) s.x1 = X[T0 + 2 * q + (p >> 1)][2 * (p & 1) + 1];
        if (k == 2) s.x0 = relu1<false>(s.x0);
        if (k == 3) s.x1 = relu1<false>(s.x1);
        if (k == 4) max3_acc(m, s.x0, s.x1);
        if (k == 5) mix_hi_lo(s.h, s.x0, r);
        if (k == 6) { mix_hi_hi(s.h, s.x1, r); o.h[p] = s.h; }
        if (k == 7) mix_lo_lo(s.l, s.x0, r, s.h);
        if (k == 8) { mix_lo_hi(s.l, s.x1, r, s.h); o.l[p] = s.l; }
    }
};
struct LdsSplit4 {
    const float* base;       // this lane's column of the parked embedding: slot s at base[s * 64]
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
    __device__ __forceinline__ void micro(int k, MicroPair& s, Split2& o, int q, int p) const {
        if (k == 0) s.x0 = base[(8 * q + 2 * p) * 64];
        if (k == 1) s.x1 = base[(8 * q + 2 * p + 1) * 64];
        if (k == 5) mix_hi_lo(s.h, s.x0, r);
        if (k == 6) { mix_hi_hi(s.h, s.x1, r); o.h[p] = s.h; }
        if (k == 7) mix_lo_lo(s.l, s.x0, r, s.h);
        if (k == 8) { mix_lo_hi(s.l, s.x1, r, s.h); o.l[p] = s.l; }
    }
};
struct BiasInit4 {
    const char* p;            // bias block + 16 g bytes
    float s;
    __device__ __forceinline__ f32x4 raw(int t) const { return *(const f32x4*)(p + t * 64); }
    __device__ __forceinline__ f32x4 operator()(int t) const {
        const f32x4 b = raw(t);
        return f32x4{b[0] * s, b[1] * s, b[2] * s, b[3] * s};
    }
};
struct ZeroInit4 {
    float s = 0.f;
    __device__ __forceinline__ f32x4 raw(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 operator()(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};

// acc[0 .. NT) (+)= W-block * src over KS steps of 32 k-values; stream order: for step q, for tile t: one 2 KiB unit.
// Block-per-pair placement and compiler-scheduled MFMAs: the SIMD's second wave fills the gaps.
template <int NT, int KS, bool FIRST, class SrcFn, class InitFn, int NACC>
__device__ __forceinline__ void run_h4_single(Ring8& ring, const char* ring_lane, const SrcFn& src, const InitFn& init, f32x4 (&acc)[NACC]) {
    static_assert(NT <= NACC, "accumulator array too small");
    constexpr int UPS = 16, NU = KS * NT, NSLAB = (NU + UPS - 1) / UPS;
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
    const char* p = ring_lane + ring.cur_off;
    f32x4 ah = ring.pf, al = *(const f32x4*)(p + 1024);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                f32x4 nh;
                const bool last = !(uu + 1 < nu);
                if (!last) {
                    nh = *(const f32x4*)(p + (2 * uu + 2) * 1024);
                } else {
#pragma unroll
                    for (int qq = 0; qq < kPieces; ++qq)
                        if ((qq * nu) / kPieces >= uu) ring.issue_piece(qq);
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off;
                    nh = *(const f32x4*)(p);
                }
                if (t == 0 && q > 0) B = Bn;
                const f16x8 Ah = as_f16x8(ah), Al = as_f16x8(al), Bh = as_f16x8(B.h), Bl = as_f16x8(B.l);
                f32x4 c = (FIRST && q == 0) ? init(t) : acc[t];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, c, 0, 0, 0);          // small terms first
                al = *(const f32x4*)(p + (last ? 1 : 2 * uu + 3) * 1024);
                if (uu + 1 < nu) {
#pragma unroll
                    for (int qq = 0; qq < kPieces; ++qq)
                        if ((qq * nu) / kPieces == uu) ring.issue_piece(qq);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, c, 0, 0, 0);
                acc[t] = c;
                if (q + 1 < KS) {
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        if (NT >= 4 ? (t == pp * (NT / 4) + NT / 4 - 1) : (t == (pp * NT) / 4)) {
                            PairRegs s1;
                            src.stage_a(s1, q + 1, pp);
                            src.stage_b(s1);
                            if (pp == 3) src.template stage_c<true>(Bn, pp, s1);
                            else src.template stage_c<false>(Bn, pp, s1);
                        }
                }
                ah = nh;
            }
        }
    }
    ring.pf = ah;
}

// The same for an even number of tiles, two units at a time: the six MFMAs of a pair of tiles alternate between the two
// accumulators (no MFMA waits for the one issued just before it) and the A operands of the NEXT pair are requested at the top
// of the pair -- twice the look-ahead of run_h4_single for 16 more registers.
template <int NT, int KS, bool FIRST, class SrcFn, class InitFn, int NACC>
__device__ __forceinline__ void run_h4_pairs(Ring8& ring, const char* ring_lane, const SrcFn& src, const InitFn& init, f32x4 (&acc)[NACC]) {
    static_assert(NT <= NACC && NT % 2 == 0, "pairs of tiles");
    constexpr int UPS = 16, NU = KS * NT, NSLAB = (NU + UPS - 1) / UPS;
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
    const char* p = ring_lane + ring.cur_off;
    f32x4 h0 = ring.pf, l0 = *(const f32x4*)(p + 1024), h1 = *(const f32x4*)(p + 2048), l1 = *(const f32x4*)(p + 3072);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;          // even
#pragma unroll
        for (int uu = 0; uu < UPS; uu += 2) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                const bool last = !(uu + 2 < nu);
                f32x4 nh0, nl0, nh1, nl1;
                if (!last) {
                    const char* pn = p + (2 * uu + 4) * 1024;
#ifdef H4_ABL_NOAREAD      // timing ablation (garbage results): what the doubled A-operand traffic of this layout costs
                    nh0 = h0; nl0 = l0; nh1 = h1; nl1 = l1;
#else
                    nh0 = *(const f32x4*)(pn); nl0 = *(const f32x4*)(pn + 1024); nh1 = *(const f32x4*)(pn + 2048); nl1 = *(const f32x4*)(pn + 3072);
#endif
                } else {
#pragma unroll
                    for (int qq = 0; qq < kPieces; ++qq)
                        if ((qq * nu) / kPieces >= uu) ring.issue_piece(qq);
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off;
                    nh0 = *(const f32x4*)(p); nl0 = *(const f32x4*)(p + 1024); nh1 = *(const f32x4*)(p + 2048); nl1 = *(const f32x4*)(p + 3072);
                }
                // the requests stay HERE, a whole pair ahead of their use: left alone, the scheduler sinks every read to just in
                // front of its MFMA to save registers, and every MFMA then waits out an LDS round trip (seen: wait 39 %)
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0 && q > 0) B = Bn;
                const f16x8 Bh = as_f16x8(B.h), Bl = as_f16x8(B.l);
                f32x4 c0 = (FIRST && q == 0) ? init(t) : acc[t], c1 = (FIRST && q == 0) ? init(t + 1) : acc[t + 1];
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(l0), Bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(l1), Bh, c1, 0, 0, 0);
                if (!last) {
#pragma unroll
                    for (int qq = 0; qq < kPieces; ++qq)
                        if ((qq * nu) / kPieces == uu) ring.issue_piece(qq);
                }
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h0), Bl, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h1), Bl, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h0), Bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h1), Bh, c1, 0, 0, 0);
                acc[t] = c0;
                acc[t + 1] = c1;
                if (q + 1 < KS) {
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        if (NT >= 8 ? (t == pp * (NT / 4) + NT / 4 - 2) : (t == 0 && (pp * NT) / 4 <= 1)) {
                            PairRegs s1;
                            src.stage_a(s1, q + 1, pp);
                            src.stage_b(s1);
                            if (pp == 3) src.template stage_c<true>(Bn, pp, s1);
                            else src.template stage_c<false>(Bn, pp, s1);
                        }
                }
                h0 = nh0; l0 = nl0; h1 = nh1; l1 = nl1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    ring.pf = h0;
}
template <int NT, int KS, bool FIRST, class SrcFn, class InitFn, int NACC>
__device__ __forceinline__ void run_h4(Ring8& ring, const char* ring_lane, const SrcFn& src, const InitFn& init, f32x4 (&acc)[NACC]) {
#ifndef H4_SINGLE
    if constexpr (NT % 2 == 0 && NT >= 8) run_h4_pairs<NT, KS, FIRST>(ring, ring_lane, src, init, acc);
    else
#endif
        run_h4_single<NT, KS, FIRST>(ring, ring_lane, src, init, acc);
}

// max over the four lanes (j, j+16, j+32, j+48) that share a sample
__device__ __forceinline__ float quad_max(float m) {
    m = pair_max(m);
    return fmaxf(m, __shfl_xor(m, 16));
}

__global__ __launch_bounds__(64 * kNW) void field_fwd_h4_sigma_kernel(H4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring_base = smem;
    float* bias_lds = (float*)(smem + 2 * 32768);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* e_lds = bias_lds + kBlobFloats + wave * (16 * 64) + lane;           // [wave][16 slots][64 lanes]
    for (int i = threadIdx.x; i < kBlobFloats; i += 64 * kNW) bias_lds[i] = a.bias[i];
    Ring8 ring;
    ring.init(a.stream, a.n_slabs, ring_base, wave, lane);
    const char* ring_lane = ring_base + lane * 16;
    const char* bias_grp = (const char*)bias_lds + 16 * g;
    const int* tab_i = (const int*)(bias_lds + kTabOff);
    const float* tab_f = bias_lds + kTabOff;
    auto wexp = [&](int seg) { return tab_i[2 * seg]; };
    auto rowb = [&](int seg) { return tab_f[2 * seg + 1]; };
    auto bmax = [&](int blk) { return tab_f[2 * kSegs + blk]; };
    ring.prime(ring_lane);
    auto tau_of = [&](float M, int ew) {
        const int t = pick_exp(M);
        return t < 100 - ew ? t : 100 - ew;
    };
    auto bias_at = [&](int off_floats, int es) { return BiasInit4{bias_grp + off_floats * 4, pow2i(es)}; };
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint32_t m_raw = (uint32_t)tile * (uint32_t)(16 * kNW) + (uint32_t)(wave * 16 + j);
        const bool ok = m_raw < (uint32_t)a.M;
        const uint32_t m = ok ? m_raw : (uint32_t)a.M - 1u;
        const uint32_t ray = a.s_magic ? __umulhi(m, a.s_magic) >> a.s_shift : m;
        const uint32_t smp = m - ray * (uint32_t)a.S;
        float x[3];
        {
            const float zz = a.z[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = add_rn(a.rays_o[ray * 3 + c], mul_rn(a.rays_d[ray * 3 + c], zz));
        }
        {   // this lane group's 16 embedding slots (h4_feature): 7 or 8 (frequency, component) pairs, sin and cos of each
            double t3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t3[c] = (double)x[c] * 0.15915494309189533577;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = g + 4 * i;
                float sn = 0.f, cs = 0.f;
                if (n < 30) {
                    const int k = n / 3, c = n - 3 * k;
                    const double tc = c == 0 ? t3[0] : (c == 1 ? t3[1] : t3[2]);
                    sincos_turns(tc, k, sn, cs);
                } else {
                    sn = g == 2 ? x[0] : x[2];
                    cs = g == 2 ? x[1] : 0.f;
                }
                e_lds[(2 * i) * 64] = sn;
                e_lds[(2 * i + 1) * 64] = cs;
            }
        }
        const float mE = fmaxf(fmaxf(1.f, fabsf(x[0])), fmaxf(fabsf(x[1]), fabsf(x[2])));
        f32x4 A[kNT], B[kNT];
        int es_a, es_b = 0;
        float M;
        {
            const int tau = tau_of(mE, wexp(S_L1));
            es_a = tau + wexp(S_L1);
            run_h4<kNT, kKSE, true>(ring, ring_lane, LdsSplit4{e_lds, pow2i(tau)}, bias_at(0, es_a), A);
            M = rowb(S_L1) * mE + bmax(0);
        }
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            const int seg1 = l1 <= 5 ? l1 - 1 : l1;            // L2 -> 1, L4 -> 3, L6 -> 6, L8 -> 8
            {
                const int ew = wexp(seg1), tau = tau_of(M, ew);
                float mx = 0.f;
                run_h4<kNT, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{A, pow2i(tau - es_a), mx}, bias_at((l1 - 1) * kW, tau + ew), B);
                M = rowb(seg1) * (quad_max(mx) * pow2i(-es_a)) + bmax(l1 - 1);
                es_b = tau + ew;
            }
            if (p == 3) break;
            {
                const int seg2 = l2 <= 4 ? l2 - 1 : (l2 == 5 ? S_L5H : l2);      // L3 -> 2, L5 -> L5H, L7 -> 7
                const int ew = wexp(seg2);
                const int tau = tau_of(p == 1 ? fmaxf(M, mE) : M, ew);
                float mx = 0.f;
                run_h4<kNT, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{B, pow2i(tau - es_b), mx}, bias_at((l2 - 1) * kW, tau + ew), A);
                if (p == 1) run_h4<kNT, kKSE, false>(ring, ring_lane, LdsSplit4{e_lds, pow2i(tau)}, ZeroInit4{}, A);
                M = rowb(seg2) * (quad_max(mx) * pow2i(-es_b)) + (p == 1 ? rowb(S_L5E) * mE : 0.f) + bmax(l2 - 1);
                es_a = tau + ew;
            }
        }
        {
            f32x4 sg[1];
            float mdummy = 0.f;
            const int tau = tau_of(M, wexp(S_SIG)), es = tau + wexp(S_SIG);
            run_h4<1, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{B, pow2i(tau - es_b), mdummy}, bias_at(8 * kW, es), sg);
            if (ok && g == 0) __builtin_nontemporal_store(softplus_ref(sg[0][0] * pow2i(-es)), &a.raw_t[(size_t)ray * a.S + smp]);
        }
    }
}

#if H4_W == 256
// The four-wave ring with HALF a slab of staging registers (layout (a) has no room for StagedRing's 32): register k carries pieces
// k and k + 4 of every slab in turn -- right behind the store of a piece the piece four places on is requested, so a load has four
// piece slots (eight units) to land instead of a whole slab.
struct Ring4 {
    const char* src;
    uint32_t n_slabs, g_next, c_slot, cur_off, my_off;   // g_next: slab the pieces k < 4 that are requested next belong to
    char* my_lds;
    f32x4 pf;
    f32x4 stage[4];
    __device__ __forceinline__ const char* piece_src(uint32_t slab, int q) const { return src + (size_t)slab * 32768 + my_off + q * 1024; }
    __device__ __forceinline__ void init(const char* stream, uint32_t nslabs, char* ring_base, int wave, int lane) {
        src = stream; n_slabs = nslabs;
        my_off = (uint32_t)(wave * 8192 + lane * 16);
        my_lds = ring_base + my_off;
        c_slot = 0; cur_off = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                     // slab 0 -> slot 0, four pieces at a time
#pragma unroll
            for (int k = 0; k < 4; ++k) stage[k] = *(const f32x4*)piece_src(0, 4 * h + k);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(my_lds + (4 * h + k) * 1024) = stage[k];
        }
        g_next = n_slabs > 1 ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) stage[k] = *(const f32x4*)piece_src(g_next, k);       // first half of slab 1 in flight
    }
    // piece q of the slab after the one being consumed: registers -> the idle slot; then request the piece four places on
    __device__ __forceinline__ void store_piece(int q) { *(f32x4*)(my_lds + (c_slot ^ 1u) * 32768 + q * 1024) = stage[q & 3]; }
    // the same as two 8-byte stores for two different MFMA gaps (a 16-byte store's 13 cycles of register-to-LDS transfer overrun a
    // 16-cycle gap on their own)
    __device__ __forceinline__ void store_half(int q, int part) {
        float2 v; v.x = stage[q & 3][2 * part]; v.y = stage[q & 3][2 * part + 1];
        *(float2*)(my_lds + (c_slot ^ 1u) * 32768 + q * 1024 + 8 * part) = v;
    }
    __device__ __forceinline__ void fetch_piece(int q) {
        if (q < 4) {
            stage[q] = *(const f32x4*)piece_src(g_next, q + 4);
        } else {
            const uint32_t nxt = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
            stage[q & 3] = *(const f32x4*)piece_src(nxt, q - 4);
            if (q == 7) g_next = nxt;
        }
    }
    __device__ __forceinline__ void issue_piece(int q) { store_piece(q); fetch_piece(q); }
    __device__ __forceinline__ uint32_t acquire() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        c_slot ^= 1u;
        return c_slot * 32768;
    }
    __device__ __forceinline__ void prime(const char* ring_lane) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur_off = 0;
        pf = *(const f32x4*)(ring_lane);
    }
};

// ================================================================================================================================
// Layout (a): FOUR waves, each lane serving TWO 16-sample halves (samples 32 w + j and 32 w + 16 + j): every A operand read from LDS
// feeds six MFMAs (three products x two halves), so the LDS traffic equals the production kernel's; the price is 2 x 2 x 64 = 256
// accumulator registers again (one wave per SIMD) and two of every per-sample scalar.  Same blob as layout (b).
template <int NT, int KS, bool FIRST, class Src0, class Src1, class Init0, class Init1, int NACC>
__device__ __forceinline__ void run_h4a_simple(Ring4& ring, const char* ring_lane, const Src0& src0, const Src1& src1, const Init0& init0,
                                        const Init1& init1, f32x4 (&acc0)[NACC], f32x4 (&acc1)[NACC]) {
    static_assert(NT <= NACC, "accumulator array too small");
    constexpr int UPS = 16, NU = KS * NT, NSLAB = (NU + UPS - 1) / UPS;
    Split2 B0, B1, Bn0, Bn1;
    auto produce = [&](auto& dst, const auto& src, int q, int pp) {
        PairRegs s1;
        src.stage_a(s1, q, pp);
        src.stage_b(s1);
        if (pp == 3) src.template stage_c<true>(dst, pp, s1);
        else src.template stage_c<false>(dst, pp, s1);
    };
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) { produce(B0, src0, 0, pp); produce(B1, src1, 0, pp); }
    Bn0 = B0; Bn1 = B1;
    const char* p = ring_lane + ring.cur_off;
    f32x4 h0 = ring.pf, l0 = *(const f32x4*)(p + 1024);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                // unit u + 1's operands are requested now (pinned: see run_h4_pairs); the slab's last unit acquires the next slab first
                // (one unit of look-ahead only: this kernel has no registers for two -- 256 accumulators + two operand pairs)
                f32x4 nh, nl;
                if (uu + 1 == nu) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES >= uu) ring.issue_piece(qq);
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off - (size_t)nu * 2048;     // unit index uu + 1 == nu now addresses the new slab
                }
                {
                    const char* pn = p + (size_t)(uu + 1) * 2048;
                    nh = *(const f32x4*)(pn);
                    nl = *(const f32x4*)(pn + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0 && q > 0) { B0 = Bn0; B1 = Bn1; }
                f32x4 c0 = (FIRST && q == 0) ? init0(t) : acc0[t], c1 = (FIRST && q == 0) ? init1(t) : acc1[t];
                const f16x8 Al = as_f16x8(l0), Ah = as_f16x8(h0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, as_f16x8(B0.h), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, as_f16x8(B1.h), c1, 0, 0, 0);
                if (uu + 1 < nu) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES == uu) ring.issue_piece(qq);
                }
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, as_f16x8(B0.l), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, as_f16x8(B1.l), c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, as_f16x8(B0.h), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, as_f16x8(B1.h), c1, 0, 0, 0);
                acc0[t] = c0;
                acc1[t] = c1;
                if (q + 1 < KS) {                                  // eight operand pairs per step (four per half)
#pragma unroll
                    for (int idx = 0; idx < 8; ++idx) {
                        const bool here = NT >= 16 ? (t == 2 * idx + 1) : (NT >= 8 ? (t == idx) : (t == (idx * NT) / 8));
                        if (here) {
                            if (idx < 4) produce(Bn0, src0, q + 1, idx);
                            else produce(Bn1, src1, q + 1, idx - 4);
                        }
                    }
                }
                h0 = nh; l0 = nl;
                if (uu + 1 == nu) p = ring_lane + ring.cur_off;       // plain addressing again from the new slab's unit 0
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    ring.pf = h0;
}

// Layout (a), 16 tiles per k-step (= one 32 KiB slab per step), with the production kernels' means (field_h3.h mma_run_h3_wide): the
// MFMAs as asm statements on AGPR accumulators, the side work placed in their six gaps per unit, A operands requested two units ahead,
// copy-free operand rotation.  An even tile t hosts operand pair t/2 of the NEXT step (pairs 0-3: half 0, 4-7: half 1: stage A behind
// the first MFMA, B behind the third, C1 behind the fifth, C2 in the odd unit that follows), the odd tile one ring piece.
__device__ __forceinline__ void mfma16_asm(f32x4& c, const f32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int KS, bool FIRST, class Src0, class Src1, class Init0, class Init1, int NACC>
__device__ __forceinline__ void run_h4a_wide(Ring4& ring, const char* ring_lane, const Src0& src0, const Src1& src1, const Init0& init0,
                                             const Init1& init1, f32x4 (&acc0)[NACC], f32x4 (&acc1)[NACC]) {
    constexpr int NT = 16, UPS = 16, NU = KS * NT;
    static_assert(NT <= NACC, "accumulator array too small");
    Split2 Ba0, Ba1, Bb0, Bb1;                                  // operands of even / odd steps, per half
    auto B0 = [&](int k) -> Split2& { return (k & 1) ? Bb0 : Ba0; };
    auto B1 = [&](int k) -> Split2& { return (k & 1) ? Bb1 : Ba1; };
    {
        PairRegs s0;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            src0.stage_a(s0, 0, pp); src0.stage_b(s0); src0.template stage_c<false>(Ba0, pp, s0);
            src1.stage_a(s0, 0, pp); src1.stage_b(s0);
            if (pp == 3) src1.template stage_c<true>(Ba1, pp, s0);
            else src1.template stage_c<false>(Ba1, pp, s0);
        }
    }
    if (FIRST) { acc0[0] = init0(0); acc1[0] = init1(0); }
    const char* p = ring_lane + ring.cur_off;
    // A operands two units ahead (three register pairs), as in the production schedule: with ONE unit of look-ahead (two pairs, no
    // spills) every unit waits out an LDS round trip and the kernel takes 13.8 ms instead of 11.6
    f32x4 ha0 = ring.pf, la0 = *(const f32x4*)(p + 1024);
    f32x4 ha1 = *(const f32x4*)(p + 2048), la1 = *(const f32x4*)(p + 3072);
    f32x4 ha2, la2;
    auto HA = [&](int k) -> f32x4& { return k % 3 == 0 ? ha0 : (k % 3 == 1 ? ha1 : ha2); };
    auto LA = [&](int k) -> f32x4& { return k % 3 == 0 ? la0 : (k % 3 == 1 ? la1 : la2); };
    PairRegs pr;                                               // one pair in flight: hosted by tile 2i, completed behind tile 2i+1's first MFMA
    MicroPair mp;
    // H4A_MICRO (default): the pair's nine instructions one per gap -- even unit gaps 1..6 = m0..m5, odd unit gaps 1, 3, 5 = m6..m8
    auto micro = [&](int k, int q1, int pp_) {
        if (pp_ < 4) src0.micro(k, mp, B0(q1), q1, pp_);
        else src1.micro(k, mp, B1(q1), q1, pp_ - 4);
    };
#pragma unroll
    for (int q = 0; q < KS; ++q) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int u = q * NT + t, uu = t;
            const bool make = q + 1 < KS, host = make && (t % 2 == 0), tail = make && (t % 2 == 1);
            const int pp = t / 2;                                  // pair hosted here / completed here
            const bool acq = uu == UPS - 2;                        // the request for unit u + 2 crosses into the next slab here
            const char* pn = p + (2 * uu + 4) * 1024;
            Split2 &Bc0 = B0(q), &Bc1 = B1(q), &Bn0 = B0(q + 1), &Bn1 = B1(q + 1);
            f32x4 &h0 = HA(u), &l0 = LA(u), &h2 = HA(u + 2), &l2 = LA(u + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (FIRST && q == 0) asm volatile("s_nop 1" : "+a"(acc0[t]), "+a"(acc1[t]));   // VALU-written C operands: two wait states
            mfma16_asm(acc0[t], l0, Bc0.h);
            __builtin_amdgcn_sched_barrier(0);
            const bool binit = FIRST && q == 0 && t + 1 < NT;      // the next tile's C operands: bias rows (shared by the halves) x 2^es
            f32x4 braw;
            if (binit) braw = init0.raw(t + 1);
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(0, q + 1, pp);
            if (tail) micro(6, q + 1, pp);
#elif !defined(H4A_ABL_NOSPLIT)
            if (host) { if (pp < 4) src0.stage_a(pr, q + 1, pp); else src1.stage_a(pr, q + 1, pp - 4); }
            if (tail) { if (pp < 4) src0.template stage_c2<false>(Bn0, pp, pr); else src1.template stage_c2<false>(Bn1, pp - 4, pr); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            mfma16_asm(acc1[t], l0, Bc1.h);
            __builtin_amdgcn_sched_barrier(0);
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(1, q + 1, pp);
#endif
#ifndef H4A_ABL_NOAREAD
            if (!acq) h2 = *(const f32x4*)(pn);
#endif
            if (binit) { acc0[t + 1][0] = braw[0] * init0.s; acc0[t + 1][1] = braw[1] * init0.s; }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(l0), "v"(Bc0.h));
            mfma16_asm(acc0[t], h0, Bc0.l);
            __builtin_amdgcn_sched_barrier(0);
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(2, q + 1, pp);
            if (tail) micro(7, q + 1, pp);
#elif !defined(H4A_ABL_NOSPLIT)
            if (host) { if (pp < 4) src0.stage_b(pr); else src1.stage_b(pr); }
#endif
            if (binit) { acc0[t + 1][2] = braw[2] * init0.s; acc0[t + 1][3] = braw[3] * init0.s; }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(Bc1.h));
            mfma16_asm(acc1[t], h0, Bc1.l);
            __builtin_amdgcn_sched_barrier(0);
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(3, q + 1, pp);
#endif
            if (binit) { acc1[t + 1][0] = braw[0] * init1.s; acc1[t + 1][1] = braw[1] * init1.s; }
            if (!host) {
#pragma unroll
                for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                    if (2 * qq + 1 == uu && uu < UPS - 2) {
#ifndef H4A_ABL_NORING
#ifndef H4A_STORE_B64      // two 8-byte stores in two gaps: tried, 21 % LDS bank conflicts (16-byte stride), slower
                        ring.store_piece(qq);
#else
                        ring.store_half(qq, 0);
#endif
#endif
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(Bc0.l));
            mfma16_asm(acc0[t], h0, Bc0.h);
            __builtin_amdgcn_sched_barrier(0);
            if (binit) { acc1[t + 1][2] = braw[2] * init1.s; acc1[t + 1][3] = braw[3] * init1.s; }
#if !defined(H4A_ABL_NORING) && defined(H4A_STORE_B64)
            if (!host) {
#pragma unroll
                for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                    if (2 * qq + 1 == uu && uu < UPS - 2) ring.store_half(qq, 1);
            }
#endif
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(4, q + 1, pp);
            if (tail) micro(8, q + 1, pp);
#elif !defined(H4A_ABL_NOSPLIT)
            if (host) { if (pp < 4) src0.stage_c1(Bn0, pp, pr); else src1.stage_c1(Bn1, pp - 4, pr); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(Bc1.l));
            mfma16_asm(acc1[t], h0, Bc1.h);
            __builtin_amdgcn_sched_barrier(0);
#if !defined(H4A_ABL_NOSPLIT) && !defined(H4A_NOMICRO)
            if (host) micro(5, q + 1, pp);
#endif
            if (!host) {
#pragma unroll
                for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                    if (2 * qq + 1 == uu && uu < UPS - 2) {
#ifndef H4A_ABL_NORING
                        ring.fetch_piece(qq);
#else
                        if (qq == 7) ring.g_next = (ring.g_next + 1 == ring.n_slabs) ? 0 : ring.g_next + 1;
#endif
                    }
            }
            if (acq) {                                             // the last piece goes in front of the acquire
#pragma unroll
                for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                    if (2 * qq + 1 >= UPS - 2) { ring.store_piece(qq); ring.fetch_piece(qq); }
                ring.cur_off = ring.acquire();
                p = ring_lane + ring.cur_off - (size_t)UPS * 2048;      // unit index uu + 2 >= 16 addresses the new slab
                pn = p + (2 * uu + 4) * 1024;
                h2 = *(const f32x4*)(pn);
            }
#ifndef H4A_ABL_NOAREAD
            l2 = *(const f32x4*)(pn + 1024);
#else
            h2 = h0; l2 = l0;
#endif
            if (uu + 1 == UPS) p = ring_lane + ring.cur_off;
            asm volatile("" ::"v"(h0), "v"(Bc0.h), "v"(Bc1.h));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    ring.pf = HA(NU);
    // the functor of the next layer reads these tiles with the vector ALU: wait states after the last MFMAs (4 passes each)
    asm volatile("s_nop 7\n\ts_nop 3" : "+a"(acc0[NT - 1]), "+a"(acc1[NT - 1]), "+a"(acc0[NT - 2]), "+a"(acc1[NT - 2]));
}

template <int NT, int KS, bool FIRST, class Src0, class Src1, class Init0, class Init1, int NACC>
__device__ __forceinline__ void run_h4a(Ring4& ring, const char* ring_lane, const Src0& src0, const Src1& src1, const Init0& init0,
                                        const Init1& init1, f32x4 (&acc0)[NACC], f32x4 (&acc1)[NACC]) {
#ifndef H4A_NOWIDE
    if constexpr (NT == 16) run_h4a_wide<KS, FIRST>(ring, ring_lane, src0, src1, init0, init1, acc0, acc1);
    else
#endif
        run_h4a_simple<NT, KS, FIRST>(ring, ring_lane, src0, src1, init0, init1, acc0, acc1);
}

__global__ __launch_bounds__(256, 1) void field_fwd_h4a_sigma_kernel(H4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring_base = smem;
    float* bias_lds = (float*)(smem + 2 * 32768);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* e_lds = bias_lds + kBlobFloats + wave * (2 * 16 * 64) + lane;       // [wave][half][16 slots][64 lanes]
    for (int i = threadIdx.x; i < kBlobFloats; i += 256) bias_lds[i] = a.bias[i];
    Ring4 ring;
    ring.init(a.stream, a.n_slabs, ring_base, wave, lane);
    const char* ring_lane = ring_base + lane * 16;
    const char* bias_grp = (const char*)bias_lds + 16 * g;
    const int* tab_i = (const int*)(bias_lds + kTabOff);
    const float* tab_f = bias_lds + kTabOff;
    auto wexp = [&](int seg) { return tab_i[2 * seg]; };
    auto rowb = [&](int seg) { return tab_f[2 * seg + 1]; };
    auto bmax = [&](int blk) { return tab_f[2 * kSegs + blk]; };
    ring.prime(ring_lane);
    auto tau_of = [&](float M, int ew) {
        const int t = pick_exp(M);
        return t < 100 - ew ? t : 100 - ew;
    };
    auto bias_at = [&](int off_floats, int es) { return BiasInit4{bias_grp + off_floats * 4, pow2i(es)}; };
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        // sample / ray indices are recomputed where they are needed (here and at the output) rather than kept alive across the tile
        auto locate = [&](int c, uint32_t& m, uint32_t& ray, uint32_t& smp) {
            const uint32_t m_raw = (uint32_t)tile * 128u + (uint32_t)(wave * 32 + 16 * c + j);
            const bool okc = m_raw < (uint32_t)a.M;
            m = okc ? m_raw : (uint32_t)a.M - 1u;
            ray = a.s_magic ? __umulhi(m, a.s_magic) >> a.s_shift : m;
            smp = m - ray * (uint32_t)a.S;
            return okc;
        };
        float mE[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t m, ray, smp;
            locate(c, m, ray, smp);
            float x[3];
            const float zz = a.z[m];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = add_rn(a.rays_o[ray * 3 + k], mul_rn(a.rays_d[ray * 3 + k], zz));
            double t3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t3[k] = (double)x[k] * 0.15915494309189533577;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = g + 4 * i;
                float sn = 0.f, cs = 0.f;
                if (n < 30) {
                    const int k = n / 3, cc = n - 3 * k;
                    sincos_turns(cc == 0 ? t3[0] : (cc == 1 ? t3[1] : t3[2]), k, sn, cs);
                } else {
                    sn = g == 2 ? x[0] : x[2];
                    cs = g == 2 ? x[1] : 0.f;
                }
                e_lds[(c * 16 + 2 * i) * 64] = sn;
                e_lds[(c * 16 + 2 * i + 1) * 64] = cs;
            }
            mE[c] = fmaxf(fmaxf(1.f, fabsf(x[0])), fmaxf(fabsf(x[1]), fabsf(x[2])));
        }
        f32x4 A0[kNT], A1[kNT], B0[kNT], B1[kNT];
        int es_a[2], es_b[2] = {0, 0};
        float M[2];
        {
            int tau[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { tau[c] = tau_of(mE[c], wexp(S_L1)); es_a[c] = tau[c] + wexp(S_L1); M[c] = rowb(S_L1) * mE[c] + bmax(0); }
            run_h4a<kNT, kKSE, true>(ring, ring_lane, LdsSplit4{e_lds, pow2i(tau[0])}, LdsSplit4{e_lds + 16 * 64, pow2i(tau[1])},
                                     bias_at(0, es_a[0]), bias_at(0, es_a[1]), A0, A1);
        }
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            const int seg1 = l1 <= 5 ? l1 - 1 : l1;
            {
                const int ew = wexp(seg1);
                int tau[2];
                float mx[2] = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) tau[c] = tau_of(M[c], ew);
                run_h4a<kNT, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{A0, pow2i(tau[0] - es_a[0]), mx[0]},
                                         ReluSplit4<kNT>{A1, pow2i(tau[1] - es_a[1]), mx[1]}, bias_at((l1 - 1) * kW, tau[0] + ew),
                                         bias_at((l1 - 1) * kW, tau[1] + ew), B0, B1);
#pragma unroll
                for (int c = 0; c < 2; ++c) { M[c] = rowb(seg1) * (quad_max(mx[c]) * pow2i(-es_a[c])) + bmax(l1 - 1); es_b[c] = tau[c] + ew; }
            }
            if (p == 3) break;
            {
                const int seg2 = l2 <= 4 ? l2 - 1 : (l2 == 5 ? S_L5H : l2);
                const int ew = wexp(seg2);
                int tau[2];
                float mx[2] = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) tau[c] = tau_of(p == 1 ? fmaxf(M[c], mE[c]) : M[c], ew);
                run_h4a<kNT, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{B0, pow2i(tau[0] - es_b[0]), mx[0]},
                                         ReluSplit4<kNT>{B1, pow2i(tau[1] - es_b[1]), mx[1]}, bias_at((l2 - 1) * kW, tau[0] + ew),
                                         bias_at((l2 - 1) * kW, tau[1] + ew), A0, A1);
                if (p == 1)
                    run_h4a<kNT, kKSE, false>(ring, ring_lane, LdsSplit4{e_lds, pow2i(tau[0])}, LdsSplit4{e_lds + 16 * 64, pow2i(tau[1])},
                                              ZeroInit4{}, ZeroInit4{}, A0, A1);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    M[c] = rowb(seg2) * (quad_max(mx[c]) * pow2i(-es_b[c])) + (p == 1 ? rowb(S_L5E) * mE[c] : 0.f) + bmax(l2 - 1);
                    es_a[c] = tau[c] + ew;
                }
            }
        }
        {
            f32x4 sg0[1], sg1[1];
            float md0 = 0.f, md1 = 0.f;
            int tau[2], es[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { tau[c] = tau_of(M[c], wexp(S_SIG)); es[c] = tau[c] + wexp(S_SIG); }
            run_h4a<1, kKSH, true>(ring, ring_lane, ReluSplit4<kNT>{B0, pow2i(tau[0] - es_b[0]), md0}, ReluSplit4<kNT>{B1, pow2i(tau[1] - es_b[1]), md1},
                                   bias_at(8 * kW, es[0]), bias_at(8 * kW, es[1]), sg0, sg1);
            if (g == 0) {
                uint32_t m, ray, smp;
                if (locate(0, m, ray, smp)) __builtin_nontemporal_store(softplus_ref(sg0[0][0] * pow2i(-es[0])), &a.raw_t[(size_t)ray * a.S + smp]);
                if (locate(1, m, ray, smp)) __builtin_nontemporal_store(softplus_ref(sg1[0][0] * pow2i(-es[1])), &a.raw_t[(size_t)ray * a.S + smp]);
            }
        }
    }
}

#endif   // layout (a): H4_W == 256 only

// ---- host side: the blob -------------------------------------------------------------------------------------------------------
struct SegDesc { int rows, real_rows, ks, layer, col0, n_cols, emb; };       // layer: index into the (weight, bias) table
const SegDesc kSeg[kSegs] = {
    {kW, kW, kKSE, 0, 0, 63, 1},  {kW, kW, kKSH, 1, 0, kW, 0}, {kW, kW, kKSH, 2, 0, kW, 0}, {kW, kW, kKSH, 3, 0, kW, 0},
    {kW, kW, kKSH, 4, 63, kW, 0}, {kW, kW, kKSE, 4, 0, 63, 1}, {kW, kW, kKSH, 5, 0, kW, 0}, {kW, kW, kKSH, 6, 0, kW, 0},
    {kW, kW, kKSH, 7, 0, kW, 0},  {16, 1, kKSH, 10, 0, kW, 0}};
int seg_slabs(const SegDesc& s) { return (s.ks * (s.rows / 16) + 15) / 16; }
int total_slabs() { int n = 0; for (const auto& s : kSeg) n += seg_slabs(s); return n; }
int ld_of(int layer) { return layer == 0 ? 63 : (layer == 4 ? 63 + kW : kW); }

uint16_t f16_bits(float f) { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
float f16_val(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

void magic_div(uint32_t d, uint32_t& magic, uint32_t& shift) {
    if (d == 1) { magic = 0; shift = 0; return; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    const uint64_t p = 31 + l;
    magic = (uint32_t)(((1ull << p) + d - 1) / d);
    shift = (uint32_t)(p - 32);
}

}  // namespace

// the H4_W = 128 object exports these three under other names; the H4_W = 256 object holds the C entry points and forwards
size_t nefes_h4_w128_blob_bytes(const NefesNetDesc* desc);
int nefes_h4_w128_pack(const NefesNetDesc* desc, const float* const* tensors, int n_tensors, void* blob, size_t blob_bytes);
int nefes_h4_w128_fwd(const NefesNetDesc* desc, const void* blob, int N, int S, const float* rays_o, const float* rays_d, const float* z,
                      float* raw_t, void* stream);
#if H4_W == 256
#define H4_FN(name) extern "C" name
#define H4_BYTES nefes_h4_sigma_blob_bytes
#define H4_PACK nefes_h4_sigma_pack
#define H4_FWD nefes_field_fwd_h4_sigma
#else
#define H4_FN(name) name
#define H4_BYTES nefes_h4_w128_blob_bytes
#define H4_PACK nefes_h4_w128_pack
#define H4_FWD nefes_h4_w128_fwd
#endif

H4_FN(size_t) H4_BYTES(const NefesNetDesc* desc) {
    if (!desc || desc->xyz_encoding != NEFES_XYZ_FREQ10) return 0;
#if H4_W == 256
    if (desc->width == 128) return nefes_h4_w128_blob_bytes(desc);
#endif
    if (desc->width != kW) return 0;
    return (size_t)kBlobFloats * 4 + (size_t)total_slabs() * 32768;
}

// tensors: the (weight, bias) table of nefes_pack_weights (xyz_encoding_1..8, xyz_encoding_final, dir_encoding, static_sigma, ...)
H4_FN(int) H4_PACK(const NefesNetDesc* desc, const float* const* tensors, int n_tensors, void* blob, size_t blob_bytes) {
#if H4_W == 256
    if (desc && desc->width == 128) return nefes_h4_w128_pack(desc, tensors, n_tensors, blob, blob_bytes);
#endif
    const size_t need = H4_BYTES(desc);
    if (!need) return NEFES_E_UNSUPPORTED;
    if (!tensors || n_tensors < 22 || !blob || blob_bytes < need) return NEFES_E_BADARG;
    memset(blob, 0, need);
    float* fl = (float*)blob;
    int* tab_i = (int*)(fl + kTabOff);
    float* tab_f = fl + kTabOff;
    for (int l = 0; l < 8; ++l) {
        const float* b = tensors[2 * l + 1];
        float mb = 0.f;
        for (int i = 0; i < kW; ++i) { fl[l * kW + i] = b[i]; mb = fmaxf(mb, fabsf(b[i])); }
        tab_f[2 * kSegs + l] = mb;
    }
    fl[8 * kW] = tensors[21][0];
    tab_f[2 * kSegs + 8] = fabsf(tensors[21][0]);
    // weight exponents: max |w| 2^e in [2^14, 2^15); the two parts of layer 5 share one (they accumulate into the same tiles)
    int wexp[kSegs];
    for (int s = 0; s < kSegs; ++s) {
        const SegDesc& sd = kSeg[s];
        const float* w = tensors[2 * sd.layer];
        const int ld = ld_of(sd.layer);
        float mx = 0.f, bound = 0.f;
        for (int r = 0; r < sd.real_rows; ++r) {
            float rs = 0.f;
            for (int c = 0; c < sd.n_cols; ++c) { const float v = fabsf(w[(size_t)r * ld + sd.col0 + c]); mx = fmaxf(mx, v); rs += v; }
            bound = fmaxf(bound, rs);
        }
        int e = 0;
        if (mx > 0.f) { int ex; frexpf(mx, &ex); e = 15 - ex; }          // mx = f 2^ex, f in [0.5, 1): mx 2^(15 - ex) in [2^14, 2^15)
        wexp[s] = e;
        tab_f[2 * s + 1] = bound;
    }
    wexp[S_L5H] = wexp[S_L5E] = wexp[S_L5H] < wexp[S_L5E] ? wexp[S_L5H] : wexp[S_L5E];
    for (int s = 0; s < kSegs; ++s) tab_i[2 * s] = wexp[s];
    char* slabs = (char*)blob + (size_t)kBlobFloats * 4;
    int slab0 = 0;
    for (int s = 0; s < kSegs; ++s) {
        const SegDesc& sd = kSeg[s];
        const float* w = tensors[2 * sd.layer];
        const int ld = ld_of(sd.layer), nt = sd.rows / 16;
        const float sc = ldexpf(1.f, wexp[s]);
        for (int q = 0; q < sd.ks; ++q)
            for (int t = 0; t < nt; ++t) {
                const int idx = q * nt + t;
                char* unit = slabs + (size_t)(slab0 + idx / 16) * 32768 + (size_t)(idx % 16) * 2048;
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4, row = 16 * t + i;
                    for (int e = 0; e < 8; ++e) {
                        int col;
                        if (sd.emb) col = h4_feature(g, 8 * q + e);
                        else col = h4_col(q, g, e);
                        float v = 0.f;
                        if (row < sd.real_rows && col >= 0 && col < sd.n_cols) v = w[(size_t)row * ld + sd.col0 + col] * sc;
                        const uint16_t hi = f16_bits(v), lo = f16_bits(v - f16_val(hi));
                        memcpy(unit + l * 16 + e * 2, &hi, 2);
                        memcpy(unit + 1024 + l * 16 + e * 2, &lo, 2);
                    }
                }
            }
        slab0 += seg_slabs(sd);
    }
    return 0;
}

H4_FN(int) H4_FWD(const NefesNetDesc* desc, const void* blob, int N, int S, const float* rays_o, const float* rays_d,
                                        const float* z, float* raw_t, void* stream) {
#if H4_W == 256
    if (desc && desc->width == 128) return nefes_h4_w128_fwd(desc, blob, N, S, rays_o, rays_d, z, raw_t, stream);
#endif
    if (!H4_BYTES(desc)) return NEFES_E_UNSUPPORTED;
    if (!blob || !rays_o || !rays_d || !z || !raw_t || N <= 0 || S <= 0) return NEFES_E_BADARG;
    H4Args a;
    a.bias = (const float*)blob;
    a.stream = (const char*)blob + (size_t)kBlobFloats * 4;
    a.n_slabs = (uint32_t)total_slabs();
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.raw_t = raw_t; a.N = N; a.S = S;
    a.M = (long long)N * S;
    if (a.M >= (1ll << 31) - 256) return NEFES_E_UNSUPPORTED;
    a.n_tiles = (int)((a.M + 16 * kNW - 1) / (16 * kNW));
    magic_div((uint32_t)S, a.s_magic, a.s_shift);
    const size_t lds = 2 * 32768 + (size_t)kBlobFloats * 4 + (size_t)kNW * 16 * 64 * 4;
#if H4_W == 256
    const char* lay = getenv("NEFES_H4_LAYOUT");                  // experiment switch: "a" = four waves x two halves, default "b"
    const bool layout_a = lay && lay[0] == 'a';
    const void* k = layout_a ? (const void*)field_fwd_h4a_sigma_kernel : (const void*)field_fwd_h4_sigma_kernel;
#else
    const bool layout_a = false;
    const void* k = (const void*)field_fwd_h4_sigma_kernel;
#endif
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = a.n_tiles < cus ? a.n_tiles : cus;
#if H4_W == 256
    if (layout_a) hipLaunchKernelGGL(field_fwd_h4a_sigma_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    else
#endif
        hipLaunchKernelGGL(field_fwd_h4_sigma_kernel, dim3(grid), dim3(64 * kNW), lds, (hipStream_t)stream, a);
    (void)layout_a;
    return (int)hipGetLastError();
}
