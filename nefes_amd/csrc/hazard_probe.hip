// What MI355X itself does with the dependencies tools/hazard_lint.py checks (round 6).  The linter's numbers are the
// wait states LLVM's GCNHazardRecognizer inserts for gfx940 / gfx950; this probe issues each producer -> consumer pair with K = 0, 1, 2, ...
// wait states between them, on every SIMD of the device, many times, and counts the lanes whose result differs from the one obtained
// with the full distance.  0 at every K = the hardware resolves the dependency itself (the rule is the toolchain's caution);
// > 0 below some K = a real software obligation, and the K at which the count reaches 0 is the measured requirement.
//
//   raw_f32_v   v_mfma_f32_32x32x2_f32 (16 passes) -> v_mov reads its VGPR result            LLVM: 18
//   raw_f16_v   v_mfma_f32_32x32x16_f16 (8 passes) -> v_mov reads its VGPR result            LLVM: 12
//   raw_f16_a   the same with the result in AGPRs, read by v_accvgpr_read_b32                LLVM: 12
//   war_b       v_mfma_f32_32x32x16_f16 reads SrcB -> v_mov overwrites that register         LLVM: none (field_h3.h holds operands by hand)
//   war_c       v_mfma_f32_32x32x16_f16 reads SrcC (other registers than vDst) -> v_mov      LLVM: 7
//   valu_b      v_mov writes SrcB -> v_mfma_f32_32x32x16_f16 reads it                        LLVM: 2
//   valu_c      v_mov writes SrcC -> v_mfma reads it                                         LLVM: 2
//   vcc_valu    v_cmp writes VCC -> v_cndmask reads it                                       LLVM: 2
//   mfma_ab     v_mfma result (VGPR) -> next v_mfma reads it as SrcB                          LLVM: 12
//   waw_v       v_mfma_f32_32x32x16_f16 result register overwritten by v_mov                 LLVM: 12
//   raw_f16_lds v_mfma_f32_32x32x16_f16 result read by ds_write_b32                          LLVM: 12
//
// Part of the library so that the driver's own GPU run records the table (tests/test_gpu_hazards.py); tools/hazard_probe.py prints it.
// tools/hazard_lint.py and tests/test_hazard_lint.py exempt these kernels by name: they exist to violate the rules.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nefes_hip.h"

// Every test: out[2 * i] = value obtained with K wait states, out[2 * i + 1] = value with the full distance.  All registers are named
// explicitly (clobbered), so that nothing the compiler does can sit between producer and consumer: one asm statement per measurement.
#define SETTLE "s_nop 15\n\ts_nop 15\n\ts_nop 7\n\t"

template <int K>
__device__ __forceinline__ void raw_f32_v(float a, float b, float& early, float& late) {
    asm volatile(
        "v_mov_b32 v32, 0\n\tv_mov_b32 v47, 0\n\t"
        ".irp r,33,34,35,36,37,38,39,40,41,42,43,44,45,46\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x2_f32 v[32:47], %2, %3, v[32:47]\n\t"
        ".if %4 > 16\n\ts_nop 15\n\ts_nop %4-17\n\t.elseif %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_mov_b32 %0, v32\n\t" SETTLE
        "v_mov_b32 %1, v32\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
}

template <int K>
__device__ __forceinline__ void raw_f16_v(unsigned a, unsigned b, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t"
        ".if %4 > 16\n\ts_nop 15\n\ts_nop %4-17\n\t.elseif %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_mov_b32 %0, v32\n\t" SETTLE
        "v_mov_b32 %1, v32\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55");
}

template <int K>
__device__ __forceinline__ void raw_f16_a(unsigned a, unsigned b, float& early, float& late) {
    asm volatile(
        ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_accvgpr_write_b32 a\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 a[0:15], v[48:51], v[52:55], a[0:15]\n\t"
        ".if %4 > 16\n\ts_nop 15\n\ts_nop %4-17\n\t.elseif %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_accvgpr_read_b32 %0, a0\n\t" SETTLE
        "v_accvgpr_read_b32 %1, a0\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(K)
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "v48", "v49", "v50", "v51", "v52",
          "v53", "v54", "v55");
}

// SrcB overwritten K wait states behind the MFMA that reads it; `late` = the same MFMA with the operand left alone
template <int K>
__device__ __forceinline__ void war_b(unsigned a, unsigned b, unsigned junk, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t"
        ".if %5 > 16\n\ts_nop 15\n\ts_nop %5-17\n\t.elseif %5 > 0\n\ts_nop %5-1\n\t.endif\n\t"
        ".irp r,52,53,54,55\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t" SETTLE
        "v_mov_b32 %0, v47\n\t"
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t" SETTLE
        "v_mov_b32 %1, v47\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(junk), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55");
}

// SrcC (v[56:71], not the destination) overwritten K wait states behind the MFMA
template <int K>
__device__ __forceinline__ void war_c(unsigned a, unsigned b, float c, float junk, float& early, float& late) {
    asm volatile(
        ".irp r,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[56:71]\n\t"
        ".if %6 > 16\n\ts_nop 15\n\ts_nop %6-17\n\t.elseif %6 > 0\n\ts_nop %6-1\n\t.endif\n\t"
        ".irp r,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71\n\tv_mov_b32 v\\r, %5\n\t.endr\n\t" SETTLE
        "v_mov_b32 %0, v47\n\t"
        ".irp r,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[56:71]\n\t" SETTLE
        "v_mov_b32 %1, v47\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "v"(junk), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
          "v70", "v71");
}

// SrcB written by the vector ALU K wait states in front of the MFMA (old value = junk, new = b)
template <int K>
__device__ __forceinline__ void valu_b(unsigned a, unsigned b, unsigned junk, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t" SETTLE
        ".irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t"
        ".if %5 > 16\n\ts_nop 15\n\ts_nop %5-17\n\t.elseif %5 > 0\n\ts_nop %5-1\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t" SETTLE
        "v_mov_b32 %0, v47\n\t"
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t" SETTLE
        "v_mov_b32 %1, v47\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(junk), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55");
}

// SrcC = vDst written by the vector ALU K wait states in front of the MFMA (old value = junk, new = c)
template <int K>
__device__ __forceinline__ void valu_c(unsigned a, unsigned b, float c, float junk, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, %5\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t"
        ".if %6 > 16\n\ts_nop 15\n\ts_nop %6-17\n\t.elseif %6 > 0\n\ts_nop %6-1\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t" SETTLE
        "v_mov_b32 %0, v47\n\t"
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, %4\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t" SETTLE
        "v_mov_b32 %1, v47\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "v"(junk), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55");
}

// VCC written by v_cmp (this lane: x > 0) K wait states in front of a v_cndmask that selects by it; VCC held the opposite before
template <int K>
__device__ __forceinline__ void vcc_valu(float x, float& early, float& late) {
    asm volatile(
        "v_cmp_lt_f32 vcc, 0, %2\n\t" SETTLE                         // vcc = (0 < x)
        "v_cmp_gt_f32 vcc, 0, %2\n\t"                                // vcc = (0 > x): flips in every lane with x != 0
        ".if %3 > 0\n\ts_nop %3-1\n\t.endif\n\t"
        "v_cndmask_b32 %0, 1.0, 2.0, vcc\n\t" SETTLE
        "v_cndmask_b32 %1, 1.0, 2.0, vcc\n\t"
        : "=&v"(early), "=&v"(late) : "v"(x), "n"(K) : "vcc");
}

// An MFMA's VGPR result as the next MFMA's SrcB, K wait states later
template <int K>
__device__ __forceinline__ void mfma_ab(unsigned a, unsigned b, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t"
        ".if %4 > 16\n\ts_nop 15\n\ts_nop %4-17\n\t.elseif %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 v[56:71], v[48:51], v[32:35], v[56:71]\n\t" SETTLE
        "v_mov_b32 %0, v71\n\t"
        ".irp r,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[56:71], v[48:51], v[32:35], v[56:71]\n\t" SETTLE
        "v_mov_b32 %1, v71\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
          "v70", "v71");
}


// XDL result register OVERWRITTEN by the vector ALU K wait states later (write after write): in program order the VALU's value stays
template <int K>
__device__ __forceinline__ void waw_v(unsigned a, unsigned b, float mark, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t"
        ".if %5 > 16\n\ts_nop 15\n\ts_nop %5-17\n\t.elseif %5 > 0\n\ts_nop %5-1\n\t.endif\n\t"
        "v_mov_b32 v32, %4\n\tv_mov_b32 v47, %4\n\t" SETTLE
        "v_add_f32 %0, v32, v47\n\t"
        "v_add_f32 %1, %4, %4\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(mark), "n"(K)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55");
}

// XDL result read by an LDS store K wait states later (ds_write_b32 of v32, read back after the settle time)
template <int K>
__device__ __forceinline__ void raw_f16_lds(unsigned a, unsigned b, unsigned lds_addr, float& early, float& late) {
    asm volatile(
        ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n\t"
        ".if %5 > 16\n\ts_nop 15\n\ts_nop %5-17\n\t.elseif %5 > 0\n\ts_nop %5-1\n\t.endif\n\t"
        "ds_write_b32 %4, v32\n\t" SETTLE
        "s_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %4\n\tv_mov_b32 %1, v32\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(lds_addr), "n"(K)
        : "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
          "v50", "v51", "v52", "v53", "v54", "v55");
}


// ---- the vector-ALU-only rules of the linter (C4, C7, C9, C11, A1 through an AGPR) ------------------------------------------------
// v_mov writes a register K wait states in front of v_permlane32_swap of it (old value = junk): LLVM 2
template <int K>
__device__ __forceinline__ void valu_swap(float x, float junk, float& early, float& late) {
    asm volatile(
        "v_mov_b32 v32, %3\n\tv_mov_b32 v33, %3\n\t" SETTLE
        "v_mov_b32 v32, %2\n\t"
        ".if %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_permlane32_swap_b32 v32, v33\n\t" SETTLE
        "v_add_f32 %0, v32, v33\n\t"
        "v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\t" SETTLE
        "v_permlane32_swap_b32 v32, v33\n\t" SETTLE
        "v_add_f32 %1, v32, v33\n\t"
        : "=&v"(early), "=&v"(late) : "v"(x), "v"(junk), "n"(K) : "v32", "v33");
}
// transcendental result read by an ordinary vector instruction K wait states later: LLVM 1
template <int K>
__device__ __forceinline__ void trans_valu(float x, float junk, float& early, float& late) {
    asm volatile(
        "v_mov_b32 v32, %3\n\t" SETTLE
        "v_exp_f32 v32, %2\n\t"
        ".if %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_add_f32 %0, v32, v32\n\t" SETTLE
        "v_add_f32 %1, v32, v32\n\t"
        : "=&v"(early), "=&v"(late) : "v"(x), "v"(junk), "n"(K) : "v32");
}
// v_mov writes a register K wait states in front of a DPP instruction that reads it: LLVM 2
template <int K>
__device__ __forceinline__ void valu_dpp(float x, float junk, float& early, float& late) {
    asm volatile(
        "v_mov_b32 v32, %3\n\tv_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\t" SETTLE
        "v_mov_b32 v32, %2\n\t"
        ".if %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_mov_b32_dpp %0, v32 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" SETTLE
        "v_mov_b32_dpp %1, v32 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(early), "=&v"(late) : "v"(x), "v"(junk), "n"(K) : "v32");
}
// v_mov writes a register K wait states in front of v_readfirstlane of it: LLVM 1
template <int K>
__device__ __forceinline__ void valu_readlane(float x, float junk, float& early, float& late) {
    asm volatile(
        "v_mov_b32 v32, %3\n\t" SETTLE
        "v_mov_b32 v32, %2\n\t"
        ".if %4 > 0\n\ts_nop %4-1\n\t.endif\n\t"
        "v_readfirstlane_b32 s40, v32\n\t" SETTLE
        "v_readfirstlane_b32 s41, v32\n\t"
        "s_nop 3\n\tv_mov_b32 %0, s40\n\tv_mov_b32 %1, s41\n\t"
        : "=&v"(early), "=&v"(late) : "v"(x), "v"(junk), "n"(K) : "v32", "s40", "s41");
}
// v_accvgpr_write of the MFMA's SrcC (an AGPR tile) K wait states in front of it (old value = junk): LLVM 2
template <int K>
__device__ __forceinline__ void accw_c(unsigned a, unsigned b, float c, float junk, float& early, float& late) {
    asm volatile(
        ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_accvgpr_write_b32 a\\r, %5\n\t.endr\n\t"
        ".irp r,48,49,50,51\n\tv_mov_b32 v\\r, %2\n\t.endr\n\t.irp r,52,53,54,55\n\tv_mov_b32 v\\r, %3\n\t.endr\n\t" SETTLE
        ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_accvgpr_write_b32 a\\r, %4\n\t.endr\n\t"
        ".if %6 > 16\n\ts_nop 15\n\ts_nop %6-17\n\t.elseif %6 > 0\n\ts_nop %6-1\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 a[0:15], v[48:51], v[52:55], a[0:15]\n\t" SETTLE
        "v_accvgpr_read_b32 %0, a15\n\t"
        ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_accvgpr_write_b32 a\\r, %4\n\t.endr\n\t" SETTLE
        "v_mfma_f32_32x32x16_f16 a[0:15], v[48:51], v[52:55], a[0:15]\n\t" SETTLE
        "v_accvgpr_read_b32 %1, a15\n\t"
        : "=&v"(early), "=&v"(late) : "v"(a), "v"(b), "v"(c), "v"(junk), "n"(K)
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "v48", "v49", "v50", "v51", "v52",
          "v53", "v54", "v55");
}

template <int TEST, int K>
__global__ __launch_bounds__(256) void hazard_probe_kernel(unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned n = 0;
    for (int it = 0; it < iters; ++it) {
        const float fa = 1.f + 0.001f * (float)(lane + it), fb = 2.f + 0.01f * (float)(it & 31);
        // two packed halves per dword, values that differ per lane and iteration (fp16 of small integers)
        const unsigned ha = 0x3c003c00u + (((lane + it) & 7) << 10), hb = 0x40004000u + (((lane * 3 + it) & 7) << 10), junk = 0x7bff7bffu;
        float e = 0.f, l = 0.f;
        if constexpr (TEST == 0) raw_f32_v<K>(fa, fb, e, l);
        if constexpr (TEST == 1) raw_f16_v<K>(ha, hb, e, l);
        if constexpr (TEST == 2) raw_f16_a<K>(ha, hb, e, l);
        if constexpr (TEST == 3) war_b<K>(ha, hb, junk, e, l);
        if constexpr (TEST == 4) war_c<K>(ha, hb, fa, 12345.f, e, l);
        if constexpr (TEST == 5) valu_b<K>(ha, hb, junk, e, l);
        if constexpr (TEST == 6) valu_c<K>(ha, hb, fa, 12345.f, e, l);
        if constexpr (TEST == 7) vcc_valu<K>((lane & 1) ? fa : -fa, e, l);
        if constexpr (TEST == 8) mfma_ab<K>(ha, hb, e, l);
        if constexpr (TEST == 9) waw_v<K>(ha, hb, 7.f + (float)lane, e, l);
        if constexpr (TEST == 10) {
            __shared__ float buf[256];
            raw_f16_lds<K>(ha, hb, (unsigned)(size_t)(&buf[threadIdx.x]), e, l);
        }
        if constexpr (TEST == 11) valu_swap<K>(fa + (float)lane, -77.f, e, l);
        if constexpr (TEST == 12) trans_valu<K>(0.01f * (float)((lane + it) & 63), 5.f, e, l);
        if constexpr (TEST == 13) valu_dpp<K>(fa + (float)lane, -77.f, e, l);
        if constexpr (TEST == 14) valu_readlane<K>(fa + (float)it, -77.f, e, l);
        if constexpr (TEST == 15) accw_c<K>(ha, hb, fa, 12345.f, e, l);
        n += (__float_as_uint(e) != __float_as_uint(l)) ? 1u : 0u;
    }
    atomicAdd(&bad[0], n);
}

template <int TEST>
static int launch_k(int k, unsigned* bad, int blocks, int iters, hipStream_t st) {
#define HZ_CASE(K) case K: hipLaunchKernelGGL((hazard_probe_kernel<TEST, K>), dim3(blocks), dim3(256), 0, st, bad, iters); break;
    switch (k) {
        HZ_CASE(0) HZ_CASE(1) HZ_CASE(2) HZ_CASE(3) HZ_CASE(4) HZ_CASE(5) HZ_CASE(6) HZ_CASE(7) HZ_CASE(8) HZ_CASE(10) HZ_CASE(12) HZ_CASE(16) HZ_CASE(18)
        default: return NEFES_E_UNSUPPORTED;
    }
#undef HZ_CASE
    return (int)hipGetLastError();
}

// `count` (device, one unsigned, zeroed by the caller) += lanes x repetitions whose result with K wait states between producer and consumer
// differs from the result with the full distance.  test: 0 raw_f32_v, 1 raw_f16_v, 2 raw_f16_a, 3 war_b, 4 war_c, 5 valu_b, 6 valu_c,
// 7 vcc_valu, 8 mfma_ab, 9 waw_v, 10 raw_f16_lds (the header comment above), 11 valu_swap, 12 trans_valu, 13 valu_dpp, 14 valu_readlane,
// 15 accw_c (vector-ALU-only rules, next to their definitions); K in {0..8, 10, 12, 16, 18}.
extern "C" int nefes_probe_hazard(int test, int k, int blocks, int iters, unsigned* count, void* stream) {
    if (!count || blocks <= 0 || iters <= 0) return NEFES_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    switch (test) {
        case 0: return launch_k<0>(k, count, blocks, iters, st);
        case 1: return launch_k<1>(k, count, blocks, iters, st);
        case 2: return launch_k<2>(k, count, blocks, iters, st);
        case 3: return launch_k<3>(k, count, blocks, iters, st);
        case 4: return launch_k<4>(k, count, blocks, iters, st);
        case 5: return launch_k<5>(k, count, blocks, iters, st);
        case 6: return launch_k<6>(k, count, blocks, iters, st);
        case 7: return launch_k<7>(k, count, blocks, iters, st);
        case 8: return launch_k<8>(k, count, blocks, iters, st);
        case 9: return launch_k<9>(k, count, blocks, iters, st);
        case 10: return launch_k<10>(k, count, blocks, iters, st);
        case 11: return launch_k<11>(k, count, blocks, iters, st);
        case 12: return launch_k<12>(k, count, blocks, iters, st);
        case 13: return launch_k<13>(k, count, blocks, iters, st);
        case 14: return launch_k<14>(k, count, blocks, iters, st);
        case 15: return launch_k<15>(k, count, blocks, iters, st);
        default: return NEFES_E_UNSUPPORTED;
    }
}
