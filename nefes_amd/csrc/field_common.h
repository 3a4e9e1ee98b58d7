// Device-side building blocks of the fused field kernels (gfx950 only).
//   - WeightRing: 8 x 16 KiB LDS ring filled by LDS-DMA (global_load_lds_dwordx4), one barrier per slab
//   - mma_segment: one (activation vector) x (weight block) product on v_mfma_f32_32x32x2_f32
//   - accumulator <-> activation-vector moves, bias init, ReLU + mask
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// exact reference arithmetic where the reference uses separate mul/add (no FMA contraction)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

// torch.nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus_ref(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.f / (1.f + expf(-x)); }

// One 1 KiB LDS-DMA piece: every lane supplies its own 16-byte source address, the destination is
// M0 (wave-uniform LDS byte address) + lane*16.  The asm statement saves/restores M0 (compiler-owned).
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// Weight stream ring.  All four waves of the workgroup walk the same slab sequence; each wave moves
// four of the sixteen 1 KiB pieces of every slab.  Protocol per consumed slab i:
//     s_waitcnt vmcnt(4*(SLOTS-2))   my pieces of slab i have landed (only younger slabs in flight)
//     s_barrier                      everyone's pieces landed; everyone is done reading slab i-1
//     issue slab i+SLOTS-1 into the slot slab i-1 occupied
// RULE (measured on gfx950: violating it gives intermittent wild-address faults): no compiler-issued global
// LOAD may be outstanding while this ring issues DMA.  hipcc counts only its own loads in the s_waitcnt it
// emits; with untracked LDS-DMA interleaved a destination VGPR can be reused before its data lands.  So the
// kernels load everything they need at the top of a tile and call loads_landed() (vmcnt(0), pinned to the
// loaded registers) before the first acquire(); everything else is LDS traffic or stores.
template <int SLOTS>
struct WeightRing {
    const char* src;     // per-lane source: stream base + wave*1024 + lane*16
    uint32_t lds_wave;   // ring base + wave*1024 (wave-uniform)
    uint32_t n_slabs;    // slabs in the stream (wraps)
    uint32_t g_next;     // next slab of the stream to issue
    uint32_t p_slot;     // slot that slab goes to
    uint32_t c_slot;     // slot consumed next
    uint32_t cur_off;    // LDS offset of the slab being consumed (already acquired)
    f32x4 pf;            // its first fragment group, prefetched

    __device__ __forceinline__ void init(const char* stream, uint32_t nslabs, uint32_t ring_lds_base, int wave, int lane) {
        src = stream + wave * 1024 + lane * 16;
        lds_wave = ring_lds_base + wave * 1024;
        n_slabs = nslabs;
        g_next = 0; p_slot = 0; c_slot = 0;
#pragma unroll 1
        for (int i = 0; i < SLOTS - 1; ++i) issue();
    }
    __device__ __forceinline__ void issue() {
        const char* s = src + (size_t)g_next * NEFES_SLAB_BYTES;
        const uint32_t d = lds_wave + p_slot * NEFES_SLAB_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma16(s + q * 4096, d + q * 4096);
        g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
        p_slot = (p_slot + 1 == SLOTS) ? 0 : p_slot + 1;
    }
    // One 1 KiB piece (q = 0..3) of the slab that refills the slot freed by the latest acquire().  The four
    // pieces are issued one at a time behind MFMAs of the slab being consumed (mma_segment), never as a burst:
    // an LDS-DMA issue costs ~60 cycles of the wave's issue slot, which one 64-cycle MFMA in flight covers.
    __device__ __forceinline__ void issue_piece(int q) {
        lds_dma16(src + (size_t)g_next * NEFES_SLAB_BYTES + q * 4096, lds_wave + p_slot * NEFES_SLAB_BYTES + q * 4096);
        if (q == 3) {
            g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
            p_slot = (p_slot + 1 == SLOTS) ? 0 : p_slot + 1;
        }
    }
    // Synchronisation point before consuming the next slab; returns its LDS byte offset (relative to the ring base).
    // After it the slot of the slab consumed before is free: its refill = the 4 issue_piece() calls that follow.
#ifdef NEFES_STAMP   // diagnostic build only: cycles parked in the counted wait and in the barrier
    unsigned long long dbg_wait = 0, dbg_barrier = 0;
    static __device__ __forceinline__ unsigned long long now() {
        unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#endif
    __device__ __forceinline__ uint32_t acquire() {
#ifdef NEFES_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SLOTS - 2)) : "memory");
        const unsigned long long t1 = now();
        __builtin_amdgcn_s_barrier();
        const unsigned long long t2 = now();
        dbg_wait += t1 - t0; dbg_barrier += t2 - t1;
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (SLOTS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
#endif
        const uint32_t off = c_slot * NEFES_SLAB_BYTES;
        c_slot = (c_slot + 1 == SLOTS) ? 0 : c_slot + 1;
        return off;
    }
    // acquire the first slab and request its first fragment group (once, before the tile loop)
    __device__ __forceinline__ void prime(const char* ring_lane) {
        cur_off = acquire();
        pf = *(const f32x4*)(ring_lane + cur_off);
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// Explicit "every outstanding vector-memory op has completed" point; pin() ties a loaded value to it so that no
// consumer (and no reuse of its register) can be scheduled above the wait.
__device__ __forceinline__ void loads_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }
template <typename T, int N>
__device__ __forceinline__ void pin(T (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// acc[t] += W-block * in   for NT accumulator tiles over KS k-steps (fully unrolled; `in` and `acc`
// are register arrays, every index below is a compile-time constant after unrolling).
// ring_lane = LDS pointer of the ring base + lane*16.
// The NT tiles are acc[T0 .. T0+NT) of a (possibly larger) accumulator array.
template <int NT, int KS, int T0 = 0, int NACC, int SLOTS>
__device__ __forceinline__ void mma_segment(WeightRing<SLOTS>& ring, const char* ring_lane, const float (&in)[KS],
                                            f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    constexpr int SPS = NEFES_SLAB_FRAGS / NT;
    constexpr int NSLAB = (KS + SPS - 1) / SPS;
    // Software pipeline: the 16-byte fragment group of the NEXT 4 MFMAs is always in flight behind the current 4.
    // At the last group of a slab the next slab (of this or of the following segment: the stream is one sequence)
    // is acquired and its first group requested BEFORE the last 4 MFMAs issue, so the barrier and the LDS latency
    // sit in the shadow of the matrix pipe.  ring.pf / ring.cur_off carry that state between segments.
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const char* p = ring_lane + ring.cur_off;
        const int steps = (KS - sl * SPS) < SPS ? (KS - sl * SPS) : SPS;
        const int nf = steps * NT;
        const int ng = (nf + 3) / 4;
        f32x4 a = ring.pf;
#pragma unroll
        for (int g = 0; g < NEFES_SLAB_FRAGS / 4; ++g) {
            if (g < ng) {
                // refill pieces scheduled in this group: piece q goes behind the first MFMA of group (q*ng)/4; in the
                // last group of the slab they go first, because all four must be issued before the next acquire()
                if (g == ng - 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((q * ng) / 4 == g) ring.issue_piece(q);
                }
                f32x4 a_next;
                if (g + 1 < ng) {
                    a_next = *(const f32x4*)(p + (g + 1) * 1024);
                } else {
                    ring.cur_off = ring.acquire();          // waits lgkmcnt(0): every read of this slab is in registers
                    a_next = *(const f32x4*)(ring_lane + ring.cur_off);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = g * 4 + q;
                    if (f < nf) {
                        const int s = sl * SPS + f / NT, t = f % NT;
                        acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], in[s], acc[T0 + t], 0, 0, 0);
                    }
                    if (q == 0 && g != ng - 1) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
                            if ((qq * ng) / 4 == g) {
                                __builtin_amdgcn_sched_barrier(0);
                                ring.issue_piece(qq);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                }
                a = a_next;
            }
        }
        ring.pf = a;
    }
}

// acc rows <- bias (natural row order in LDS).  bias_half = bias block + 16*h bytes.
template <int NT>
__device__ __forceinline__ void bias_init(f32x16 (&acc)[NT], const char* bias_half) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4*)(bias_half + (t * 32 + 8 * q) * 4);
            acc[t][4 * q + 0] = b[0]; acc[t][4 * q + 1] = b[1]; acc[t][4 * q + 2] = b[2]; acc[t][4 * q + 3] = b[3];
        }
}
template <int NT, int T0 = 0, int NACC>
__device__ __forceinline__ void zero_init(f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[T0 + t][r] = 0.f;
}

// dst[16*t + r] = max(acc[t][r], floor);  mask bit (16*t + r) = acc > 0.   floor = 0 (ReLU) or -inf (identity)
template <int NT, int NOUT>
__device__ __forceinline__ void act_store(float (&dst)[NOUT], const f32x16 (&acc)[NT], float floor_v,
                                          uint32_t (&bits)[(NT + 1) / 2]) {
    static_assert(NOUT >= NT * 16, "destination vector too small");
#pragma unroll
    for (int w = 0; w < (NT + 1) / 2; ++w) bits[w] = 0u;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[t][r];
            bits[t >> 1] |= (v > 0.f ? 1u : 0u) << ((t & 1) * 16 + r);
            dst[t * 16 + r] = fmaxf(v, floor_v);
        }
}
// backward: dst[16*t + r] = bit ? acc[T0 + t][r] : 0
template <int NT, int T0, int NACC, int NOUT>
__device__ __forceinline__ void mask_store(float (&dst)[NOUT], const f32x16 (&acc)[NACC], const uint32_t (&bits)[(NT + 1) / 2]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dst[t * 16 + r] = ((bits[t >> 1] >> ((t & 1) * 16 + r)) & 1u) ? acc[T0 + t][r] : 0.f;
}

// frequency embedding of a 3-vector into slot order (layout.h nefes_emb_slot); L frequencies.
// half-0 lanes keep sin, half-1 lanes keep cos.  torch: sin(x * 2^k), cos(x * 2^k); x*2^k is exact.
template <int L, int NS>
__device__ __forceinline__ void embed_slots(float (&e)[NS], const float (&x)[3], int h) {
    static_assert(NS >= 3 * L + 2, "embedding vector too small");
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sn, cs;
            sincosf(x[a] * (float)(1 << k), &sn, &cs);
            e[3 * k + a] = h ? cs : sn;
        }
    e[3 * L] = h ? x[1] : x[0];
    e[3 * L + 1] = h ? 0.f : x[2];
#pragma unroll
    for (int s = 3 * L + 2; s < NS; ++s) e[s] = 0.f;
}
// backward of embed_slots: g[s] holds d/d(slot (s, h)); returns this lane-half's partial d/dx
// (the caller adds the other half's).   d sin(fx)/dx = f cos(fx),  d cos(fx)/dx = -f sin(fx).
template <int L, int NS>
__device__ __forceinline__ void embed_slots_bwd(float (&gx)[3], const float (&g)[NS], const float (&x)[3], int h) {
    gx[0] = gx[1] = gx[2] = 0.f;
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sn, cs;
            const float f = (float)(1 << k);
            sincosf(x[a] * f, &sn, &cs);
            gx[a] += g[3 * k + a] * (h ? -(f * sn) : (f * cs));
        }
    if (h) { gx[1] += g[3 * L]; } else { gx[0] += g[3 * L]; gx[2] += g[3 * L + 1]; }
}
