// Device-side building blocks of the fused field kernels (gfx950 only).
//   - WeightRing: LDS ring of NEFES_SLAB_KIB-KiB slabs (4 x 32 forward, 3 x 32 backward) filled by LDS-DMA
//     (global_load_lds_dwordx4), one barrier per slab
//   - mma_segment: one (activation vector) x (weight block) product on v_mfma_f32_32x32x2_f32
//   - accumulator <-> activation-vector moves, bias init, ReLU + mask
#pragma once
#ifndef NEFES_B_BATCH
#define NEFES_B_BATCH 4      /* k-steps whose B operands are produced in one VALU gap (mma_run) */
#endif
#ifndef NEFES_B_BATCH_NT8
#define NEFES_B_BATCH_NT8 NEFES_B_BATCH   /* same, for 8-tile (Wd = 256 trunk) segments; a kernel TU may raise it */
#endif
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
// Source address = wave-uniform 64-bit base (SGPR pair, advanced by scalar ALU) + a constant 32-bit per-lane offset:
// the refill costs no vector ALU instruction (VALU ops are never hidden behind an fp32 MFMA of the same wave).
__device__ __forceinline__ void lds_dma16(const void* gbase, uint32_t lane_off, uint32_t lds_dst) {
    uint32_t keep;
    // The SGPR base may have just been written by a VALU instruction (v_readlane restoring a spilled SGPR,
    // v_readfirstlane), and a VMEM instruction reading such an SGPR needs 5 wait states the compiler does not insert for
    // inline asm.  Instead of padding (s_nop 4 = 20 cycles, a large part of a 32-cycle bf16 MFMA gap) the base is copied
    // by the scalar ALU into a fresh pair that the DMA reads: SALU-written SGPRs carry no such hazard.
    // M0 is left holding the LDS address rather than saved and restored.  It cannot be declared clobbered: M0 is a reserved
    // register for this target and hipcc rejects the promise ("inline asm clobber list contains reserved registers: m0 ...
    // may not be preserved"), so soundness rests on the compiler using M0 for nothing else in these kernels -- which
    // tests/test_pack_stream.py::test_m0_only_written_by_the_dma_helper verifies on the disassembly of the shipped library
    // (and which fails the CPU suite, not silently miscompiles, should a new hipcc start to).  A save/restore would have
    // to wait until the DMA has read M0
    // (tools/probe/overlap_probe.hip, dma variant 2: one cycle per bf16 MFMA less in the x6 instruction mix).
    uint64_t base2;
    (void)keep;
    asm volatile(
        "s_mov_b64 %0, %2\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %0"
        : "=&s"(base2)
        : "v"(lane_off), "s"(gbase), "s"(lds_dst)
        : "memory");
}
// 1-instruction ReLU (fmaxf() costs a canonicalising v_max in front of the real one)
// NOP = true: the result feeds an MFMA as srcB straight away.  A VALU write -> MFMA operand read needs 2 wait states that
// the compiler does not pad for values defined inside inline asm (measured: without them the MFMA consumes the previous
// k-step's operand), hence the trailing s_nop 1 (8 cycles of issue).  NOP = false: the caller guarantees at least one
// MFMA issues between this instruction and the consumer (mma_run computes the operand one k-step ahead).
template <bool NOP = true>
__device__ __forceinline__ float relu1(float v) {
    float r;
    if (NOP) asm volatile("v_max_f32 %0, 0, %1\n\ts_nop 1" : "=v"(r) : "v"(v));
    else asm volatile("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// ReLU-mask words: forward shifts the SIGN bit of the pre-activation in from the right (v_alignbit_b32: {word, v} >> 31 = word<<1 |
// sign), one VALU instruction per activation; 32 activations per word, walked in the same order on both sides, so activation e of a
// layer ends up in bit 31 - (e % 32) of word e / 32.  Bit = 1 means "pre-activation negative" (relu' = 0).  Backward picks that bit
// (v_bfe_i32: 0 or -1) and clears the gradient where it is set (v_bfi_b32): two VALU instructions per value, none of them through
// an SGPR.  (Rounds 1-5 shifted the bits out through the carry: v_add_co_u32 vcc + v_cndmask_b32 back to back inside one asm
// statement -- two wait states short of what hipcc itself puts between a VALU write of VCC and a VALU read of it as a scalar operand
// on gfx940+, and which it cannot see inside asm: tools/hazard_lint.py rule C10, DESIGN.md section 4.10.  No wrong result was ever
// traced to it; the pick costs the same two instructions and drops the serial chain through the mask word.)
// (A pre-activation of exactly +0.0 passes the gradient where torch's relu' gives 0; -0.0 does not.)
__device__ __forceinline__ void mask_shift_in(uint32_t& bits, float v) {
    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(bits) : "v"(v));
}
// e = index of the activation in its layer's walking order (a compile-time constant after unrolling); bits = word e / 32
template <bool NOP = true>
__device__ __forceinline__ float mask_pick(const uint32_t& bits, int e, float v) {
    float r;                    // (also the scratch for the picked bit: early-clobber, it is written before v is read)
    if (NOP) asm volatile("v_bfe_i32 %0, %1, %2, 1\n\tv_bfi_b32 %0, %0, 0, %3\n\ts_nop 1" : "=&v"(r) : "v"(bits), "n"(31 - (e & 31)), "v"(v));
    else asm volatile("v_bfe_i32 %0, %1, %2, 1\n\tv_bfi_b32 %0, %0, 0, %3" : "=&v"(r) : "v"(bits), "n"(31 - (e & 31)), "v"(v));
    return r;
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
    const char* src;     // stream base (wave-uniform)
    uint32_t lane_off;   // wave*1024 + lane*16
    uint32_t lds_wave;   // ring base + wave*1024 (wave-uniform)
    uint32_t n_slabs;    // slabs in the stream (wraps)
    uint32_t g_next;     // next slab of the stream to issue
    uint32_t p_slot;     // slot that slab goes to
    uint32_t c_slot;     // slot consumed next
    uint32_t cur_off;    // LDS offset of the slab being consumed (already acquired)
    f32x4 pf;            // its first fragment group, prefetched

    __device__ __forceinline__ void init(const char* stream, uint32_t nslabs, uint32_t ring_lds_base, int wave, int lane) {
        src = stream;
        lane_off = (uint32_t)(wave * 1024 + lane * 16);
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
        for (int q = 0; q < NEFES_SLAB_PIECES; ++q) lds_dma16(s + q * 4096, lane_off, d + q * 4096);
        g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
        p_slot = (p_slot + 1 == SLOTS) ? 0 : p_slot + 1;
    }
    // One 1 KiB piece (q = 0..3) of the slab that refills the slot freed by the latest acquire().  The four
    // pieces are issued one at a time behind MFMAs of the slab being consumed (mma_segment), never as a burst:
    // an LDS-DMA issue costs ~60 cycles of the wave's issue slot, which one 64-cycle MFMA in flight covers.
    __device__ __forceinline__ void issue_piece(int q) {
        lds_dma16(src + (size_t)g_next * NEFES_SLAB_BYTES + q * 4096, lane_off, lds_wave + p_slot * NEFES_SLAB_BYTES + q * 4096);
        if (q == NEFES_SLAB_PIECES - 1) {
            g_next = (g_next + 1 == n_slabs) ? 0 : g_next + 1;
            p_slot = (p_slot + 1 == SLOTS) ? 0 : p_slot + 1;
        }
    }
    // Synchronisation point before consuming the next slab; returns its LDS byte offset (relative to the ring base).
    // After it the slot of the slab consumed before is free: its refill = the 4 issue_piece() calls that follow.
    __device__ __forceinline__ uint32_t acquire() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEFES_SLAB_PIECES * (SLOTS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
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

// ---- results of compiler-placed MFMAs that ASM statements read -----------------------------------------------------------
// hipcc pads the wait states between an MFMA and a vector instruction that touches its result (12 after the 8-pass
// v_mfma_f32_32x32x16_f16 / bf16, 18 after the 16-pass v_mfma_f32_32x32x2_f32 on gfx950) -- for vector instructions it can see.  An
// `asm volatile` statement is not one of them (GCNHazardRecognizer: INLINEASM is neither VALU nor MFMA).  Where the functors of the
// NEXT product read a run's tiles through asm -- directly out of the MFMA's own VGPRs in the objects built with
// -amdgpu-mfma-vgpr-form (the Wd = 128 fp16 instances), or through field_h3.h acc_read's asm v_accvgpr_read_b32 (the Wd = 256
// inference objects) -- nothing stood between the run's last MFMA and that read but whatever the scheduler had put there:
// tools/hazard_lint.py found the transient heads' fp32 product read 2-6 wait states after its last k-step in every such backward
// instance of round 5's library, the headline one included.  Measured on MI355X (csrc/hazard_probe.hip): the
// fp32 MFMA's result is interlocked (right at zero wait states), so no wrong number ever came of THAT read; a 16-bit (XDL) MFMA's
// result is not -- a vector read within four wait states returns the old register contents, a vector write within seven is lost --
// and the bf16x6 forward did have such a read (four wait states, on the short side of a branch hipcc's recognizer does not follow).
// The numbers used here are the toolchain's own (12 / 18: what hipcc inserts for its own code), a superset of the measured ones; the
// library now satisfies them everywhere, at no measurable cost, and the linter keeps it so.
// Translation units that read accumulators through asm define NEFES_ASM_READS_ACC; their runs end with this fence.
template <int WS>
__device__ __forceinline__ void mfma_results_fence() {
#if defined(NEFES_ASM_READS_ACC) && !defined(NEFES_NO_RESULTS_FENCE)
    static_assert(WS >= 1 && WS <= 32, "one or two s_nop");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (WS > 16) asm volatile("s_nop 15\n\ts_nop %0" ::"n"(WS - 17) : "memory");
    else asm volatile("s_nop %0" ::"n"(WS - 1) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// The same wait states TIED TO THE RUN'S TILES instead of fenced off with scheduling barriers (the fp16 objects: NEFES_FENCE_TILES):
// every MFMA of the run precedes the statement (it reads and writes all NT tiles), every reader of a tile follows it, and everything
// else -- the next run's LDS reads, mask words, exponent arithmetic -- moves across it freely.  (With sched_barrier(0) on both sides
// the Wd = 128 kernels lost 7-11 %: profiles/r06/README.md; the wait states themselves are 0.6 % of a tile.)
// NEFES_ACC_VGPR_FORM: the object is built with -amdgpu-mfma-vgpr-form, its accumulators are VGPRs.
#if defined(NEFES_ACC_VGPR_FORM)
#define NEFES_ACC_RW(x) "+v"(x)
#else
#define NEFES_ACC_RW(x) "+a"(x)
#endif
template <int WS, int NT, int T0, int NACC>
__device__ __forceinline__ void mfma_results_fence_tiles(f32x16 (&acc)[NACC]) {
#if defined(NEFES_ASM_READS_ACC) && !defined(NEFES_NO_RESULTS_FENCE)
#if !defined(NEFES_FENCE_TILES)
    mfma_results_fence<WS>();
#else
    static_assert(WS == 12 || WS == 18, "12: 8-pass XDL MFMAs, 18: the 16-pass fp32 MFMA");
    static_assert(NT >= 1 && NT <= 10, "tiles of one run");
#define NEFES_FENCE_ASM(...)                                                                  \
    do {                                                                                      \
        if constexpr (WS == 12) asm volatile("s_nop 11" : __VA_ARGS__);                       \
        else asm volatile("s_nop 15\n\ts_nop 1" : __VA_ARGS__);                                \
    } while (0)
#define A_(i) NEFES_ACC_RW(acc[T0 + (i)])
    if constexpr (NT == 1) NEFES_FENCE_ASM(A_(0));
    else if constexpr (NT == 2) NEFES_FENCE_ASM(A_(0), A_(1));
    else if constexpr (NT == 3) NEFES_FENCE_ASM(A_(0), A_(1), A_(2));
    else if constexpr (NT == 4) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3));
    else if constexpr (NT == 5) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4));
    else if constexpr (NT == 6) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4), A_(5));
    else if constexpr (NT == 7) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4), A_(5), A_(6));
    else if constexpr (NT == 8) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4), A_(5), A_(6), A_(7));
    else if constexpr (NT == 9) NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4), A_(5), A_(6), A_(7), A_(8));
    else NEFES_FENCE_ASM(A_(0), A_(1), A_(2), A_(3), A_(4), A_(5), A_(6), A_(7), A_(8), A_(9));
#undef A_
#undef NEFES_FENCE_ASM
#endif
#endif
}

// ---- B-operand producers (one float per k-step and lane) and C-operand initialisers (one tile at a time) ----------
// Consumer-side activation: a layer never materialises its activated output.  The NEXT layer's product reads the
// producer's accumulators register by register (k-step s <-> accumulator tile s/16, register s%16), applies the
// activation on the way in and, in the forward pass, records the 1-bit ReLU mask.  The two or three VALU ops per k-step
// issue in the shadow of that k-step's NT MFMAs instead of as a serial epilogue between layers.
template <int NX, int NW>
struct ReluCapture {            // forward: relu(X) and mask bit s
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NW];
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const {
        const float v = X[s >> 4][s & 15];
        mask_shift_in(bits[s >> 5], v);
        return relu1<NOP>(v);
    }
};
template <int NX>
struct ReluIn {                 // forward: relu(X), no mask
    const f32x16 (&X)[NX];
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const { return relu1<NOP>(X[s >> 4][s & 15]); }
};
template <int NX, int T0 = 0>
struct IdentIn {                // X as is (tiles T0..)
    const f32x16 (&X)[NX];
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const { return X[T0 + (s >> 4)][s & 15]; }
};
template <int NX, int NW, int T0 = 0>
struct MaskedIn {               // backward: mask bit of activation s ? 0 : X
    const f32x16 (&X)[NX];
    uint32_t (&bits)[NW];
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const { return mask_pick<NOP>(bits[s >> 5], s, X[T0 + (s >> 4)][s & 15]); }
};
// TRAIN instances of the backward kernel: the (masked) gradient vector a product consumes is also what the weight-gradient
// kernels need, so it is stored on the way in -- element s (feature 32*(s/16) + rho_h(s%16)) of this lane's sample goes to
// row p[...] of the tile-major gradient buffer (layout.h); p already holds the block's first row, the lane half's +4 rows
// and the sample column.
template <class Inner>
struct Storing {
    Inner in;
    float* p;
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const {
        const float v = in.template get<NOP>(s);
        p[(s >> 4) * 4096 + nefes_rho(0, s & 15) * 16] = v;      // layout.h nefes_train_off
        return v;
    }
};
template <bool ON, class Inner>
__device__ __forceinline__ auto wrap_store(const Inner& in, float* p) {
    if constexpr (ON) return Storing<Inner>{in, p};
    else return in;
}

template <int N>
struct ArrayIn {
    const float (&v)[N];
    template <bool NOP>
    __device__ __forceinline__ float get(int s) const { return v[s]; }
};
struct ZeroInit {               // C operand of the first k-step = 0
    __device__ __forceinline__ f32x16 operator()(int) const {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        return z;
    }
};
struct BiasInit {               // C operand of the first k-step = bias rows of tile t (natural order in LDS)
    const char* p;              // bias block of the layer + 16*h bytes
    __device__ __forceinline__ f32x16 operator()(int t) const {
        f32x16 c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4*)(p + (t * 32 + 8 * q) * 4);
            c[4 * q + 0] = b[0]; c[4 * q + 1] = b[1]; c[4 * q + 2] = b[2]; c[4 * q + 3] = b[3];
        }
        return c;
    }
};

// acc[T0 .. T0+NT) (+)= W-block * in   over KS k-steps, fully unrolled: every register index below is a compile-time
// constant after unrolling.  FIRST: the first k-step takes its C operand from init(t) (bias or zero) instead of acc,
// so accumulators need no initialisation pass.  ring_lane = LDS pointer of the ring base + lane*16.
// FENCE = false: the caller's next statement provides the wait states itself (field_h3.h H3_WIDE_ENTRY_FENCE)
template <int NT, int KS, int T0, bool FIRST, bool FENCE = true, class InFn, class InitFn, int NACC, class Ring>
__device__ __forceinline__ void mma_run(Ring& ring, const char* ring_lane, const InFn& in, const InitFn& init,
                                        f32x16 (&acc)[NACC]) {
    static_assert(T0 + NT <= NACC, "accumulator array too small");
    constexpr int SPS = NEFES_SLAB_FRAGS / NT;
    constexpr int NSLAB = (KS + SPS - 1) / SPS;
    // Software pipeline: the 16-byte fragment group of the NEXT 4 MFMAs is always in flight behind the current 4.
    // At the last group of a slab the next slab (of this or of the following segment: the stream is one sequence)
    // is acquired and its first group requested BEFORE the last 4 MFMAs issue, so the barrier and the LDS latency
    // sit in the shadow of the matrix pipe.  ring.pf / ring.cur_off carry that state between segments.
    f32x16 c0, c1;              // C operands of the first k-step, fetched two tiles ahead
    if (FIRST) {
        c0 = init(0);
        if (NT > 1) c1 = init(1);
    }
    // B operands are produced AHEAD of their consumer, at least two MFMAs earlier: the VALU instructions then need no
    // wait-state padding (2 wait states = 2 issued instructions; with only one MFMA in between the parity tests fail).
    // NT >= 3: in front of the last two MFMAs of the previous k-step (right behind its FIRST MFMA measured 2.5 % slower: a
    // VALU instruction waits for the fp32 MFMA in flight, which pushed the LDS-DMA issue that follows out of its shadow);
    // NT = 2: two k-steps earlier; NT = 1: three.  And in BATCHES of NEFES_B_BATCH k-steps: the first VALU instruction
    // after an MFMA costs ~21 cycles, each further one 4 (tools/probe/overlap_probe.hip), so one VALU gap per batch.
    constexpr int BATCH = (NT == 8) ? NEFES_B_BATCH_NT8 : NEFES_B_BATCH;
    constexpr int LA = NT >= 3 ? 1 : (NT == 2 ? 2 : 3);      // look-ahead in k-steps
    constexpr int PROD_T = NT >= 3 ? NT - 3 : 0;              // tile index behind whose MFMA a batch is produced
    constexpr int QD = 2 * BATCH;                             // queue depth: a batch is consumed before it is overwritten
    static_assert(LA <= BATCH, "look-ahead must not exceed the batch size");
    float b = 0.f, bq[QD];
#pragma unroll
    for (int i = 0; i < QD; ++i) bq[i] = 0.f;
    // the first LA operands up front, in k order (the mask-capturing producers shift bits in call order)
#pragma unroll
    for (int i = 0; i < LA; ++i)
        if (i < KS) bq[i] = in.template get<true>(i);
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
                    for (int q = 0; q < NEFES_SLAB_PIECES; ++q)
                        if ((q * ng) / NEFES_SLAB_PIECES == g) ring.issue_piece(q);
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
                        if (t == 0) {
                            // keep the MFMAs issued since the operand was produced in front of its consumer (the
                            // scheduler may otherwise hoist this MFMA: seen with NT = 9, where k-steps straddle groups)
                            __builtin_amdgcn_sched_barrier(0);
                            b = bq[s % QD];
                        }
                        if (FIRST && s == 0) {
                            acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b, c0, 0, 0, 0);
                            c0 = c1;
                            if (t + 2 < NT) c1 = init(t + 2);
                        } else {
                            acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b, acc[T0 + t], 0, 0, 0);
                        }
                        if (t == PROD_T && s % BATCH == 0 && s + LA < KS) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < BATCH; ++i)
                                if (s + LA + i < KS) bq[(s + LA + i) % QD] = in.template get<false>(s + LA + i);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (q == 0 && g != ng - 1) {
#pragma unroll
                        for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                            if ((qq * ng) / NEFES_SLAB_PIECES == g) {
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
    if constexpr (FENCE) mfma_results_fence_tiles<18, NT, T0>(acc);
}
// array-input, accumulate-into-acc form (callers initialise acc themselves)
template <int NT, int KS, int T0 = 0, int NACC, class Ring>
__device__ __forceinline__ void mma_segment(Ring& ring, const char* ring_lane, const float (&in)[KS],
                                            f32x16 (&acc)[NACC]) {
    mma_run<NT, KS, T0, false>(ring, ring_lane, ArrayIn<KS>{in}, ZeroInit{}, acc);
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
// backward: dst[s] = (mask bit of activation s) ? acc[T0 + s/16][s%16] : 0, words consumed in shift order (see mask_shift_in)
template <int NT, int T0, int NACC, int NOUT>
__device__ __forceinline__ void mask_store(float (&dst)[NOUT], const f32x16 (&acc)[NACC], const uint32_t (&bits)[(NT + 1) / 2]) {
#pragma unroll
    for (int s = 0; s < NT * 16; ++s) dst[s] = mask_pick(bits[s >> 5], s, acc[T0 + (s >> 4)][s & 15]);
}

// sin/cos of x*2^k for the frequency embedding.  x*2^k is exact in fp32 (the reference computes sin(x * 2^k)), so the argument
// reduction can be done exactly in turns -- and in integers: t = x/(2 pi) in f64 (|error| < 1e-13 turns at |x*2^k| = 2048), its
// fraction as a 64-bit fixed-point number hi.lo (once per coordinate: the only f64 instructions left, they issue at a quarter
// of the fp32 rate), and frac(t 2^k) is that number shifted left by k: one v_alignbit_b32 per frequency for the top 32 bits.
// Read as a signed integer the phase is already reduced to [-1/2, 1/2) turns; cos is sin a quarter turn later (an integer add),
// so every lane evaluates ONE odd polynomial on the half turn folded onto [-1/4, 1/4] (sin(pi - th) = sin th): twelve fp32-rate
// instructions per value instead of ~26 + four f64 (two polynomials, quadrant selects, f64 ldexp / rint / subtract / convert).
// Fit of sin(2 pi r) / r in r^2 on [0, 1/4]: 3e-9; evaluated in fp32: 1.7e-7 abs, like a 1-2 ulp libm.  VALU count matters:
// the embedding of a tile is not overlapped with that wave's MFMAs.
__device__ __forceinline__ void turns_fixed(float x, uint32_t& hi, uint32_t& lo) {
    const double t = (double)x * 0.15915494309189533577;
    const double s = (t - __builtin_floor(t)) * 4294967296.0;      // frac(t) 2^32: [0, 2^32], exact scaling
    hi = (uint32_t)s;                                              // truncating, saturating conversions
    lo = (uint32_t)((s - (double)hi) * 4294967296.0);
}
__device__ __forceinline__ float sin_phase(uint32_t top) {         // sin(2 pi top / 2^32)
    const float r = (float)(int32_t)top * 2.3283064365386963e-10f; // [-1/2, 1/2] turns
    const float a = fabsf(r);
    const float rr = __builtin_copysignf(fminf(a, 0.5f - a), r);   // folded onto [-1/4, 1/4]; 0.5 - a is exact where it is the smaller
    const float z = rr * rr;
    float p = 39.53672409057617f;
    p = __builtin_fmaf(p, z, -76.5497817993164f);
    p = __builtin_fmaf(p, z, 81.60100555419922f);
    p = __builtin_fmaf(p, z, -41.34165573120117f);
    p = __builtin_fmaf(p, z, 6.283185005187988f);
    return rr * p;
}
__device__ __forceinline__ uint32_t phase_of(uint32_t hi, uint32_t lo, int k) {   // top 32 bits of frac(t 2^k), k < 32 (compile-time)
    return k ? __builtin_amdgcn_alignbit(hi, lo, 32 - k) : hi;
}
// the round-1 form (f64 reduction per value, both polynomials): used by the experimental kernels of field_fwd_h4.hip and, with
// -DNEFES_SINCOS_F64, by embed_slots / embed_slots_bwd for A/B timing
__device__ __forceinline__ void sincos_turns(double t, int k, float& sn, float& cs) {
    const double tk = __builtin_ldexp(t, k);
    const float r = (float)(tk - __builtin_rint(tk));          // [-0.5, 0.5] turns, exact difference
    const float q = __builtin_rintf(r * 4.f);                  // quadrant -2..2
    const float th = __builtin_fmaf(-0.25f, q, r) * 6.28318530717958647692f;   // [-pi/4, pi/4]
    const float z = th * th;
    const float ps = th + th * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float pc = 1.f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int qi = (int)q & 3;
    const float s0 = (qi & 1) ? pc : ps, c0 = (qi & 1) ? ps : pc;
    sn = (qi & 2) ? -s0 : s0;
    cs = ((qi + 1) & 2) ? -c0 : c0;
}

// frequency embedding of a 3-vector into slot order (layout.h nefes_emb_slot); L frequencies.
// half-0 lanes keep sin, half-1 lanes keep cos.  torch: sin(x * 2^k), cos(x * 2^k); x*2^k is exact.
template <int L, int NS>
__device__ __forceinline__ void embed_slots(float (&e)[NS], const float (&x)[3], int h) {
    static_assert(NS >= 3 * L + 2, "embedding vector too small");
    uint32_t hi[3], lo[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) turns_fixed(x[a], hi[a], lo[a]);
    const uint32_t quarter = h ? 0x40000000u : 0u;                  // half-1 lanes keep cos = sin a quarter turn later
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) e[3 * k + a] = sin_phase(phase_of(hi[a], lo[a], k) + quarter);
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
    uint32_t hi[3], lo[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) turns_fixed(x[a], hi[a], lo[a]);
    const uint32_t shift = h ? 0x80000000u : 0x40000000u;           // d sin = cos = sin(. + 1/4 turn); d cos = -sin = sin(. + 1/2 turn)
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float f = (float)(1 << k);
            gx[a] += g[3 * k + a] * (f * sin_phase(phase_of(hi[a], lo[a], k) + shift));
        }
    if (h) { gx[1] += g[3 * L]; } else { gx[0] += g[3 * L]; gx[2] += g[3 * L + 1]; }
}
