// Issue cost of the vector instructions the Wd = 128 field kernels could build their operand split from (those
// kernels are bound by the vector issue port -- 5.5 vector instructions per MFMA, eight of every seventeen being v_fma_mix*):
// s_memtime cycles per instruction over an unrolled stream of independent instructions, with one wave per SIMD (256 threads per CU)
// and with two (512).  hipcc --offload-arch=gfx950 -O2 tools/probe/valu_probe.hip -o tools/probe/valu_probe && tools/probe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KINDS(X)                                                                                                                          \
    X(0, "v_fma_f32", "v_fma_f32 %0, %4, %5, %0", "v_fma_f32 %1, %4, %5, %1", "v_fma_f32 %2, %4, %5, %2", "v_fma_f32 %3, %4, %5, %3")      \
    X(1, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]", "v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]",          \
      "v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]", "v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]")                                \
    X(2, "v_fma_mix_f32 (f16 srcC)", "v_fma_mix_f32 %0, %4, %5, -%6 op_sel_hi:[0,0,1]", "v_fma_mix_f32 %1, %4, %5, -%6 op_sel_hi:[0,0,1]", \
      "v_fma_mix_f32 %2, %4, %5, -%6 op_sel_hi:[0,0,1]", "v_fma_mix_f32 %3, %4, %5, -%6 op_sel_hi:[0,0,1]")                                \
    X(3, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %4, %5", "v_cvt_pk_f16_f32 %1, %4, %5", "v_cvt_pk_f16_f32 %2, %4, %5",                  \
      "v_cvt_pk_f16_f32 %3, %4, %5")                                                                                                      \
    X(4, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %4, %5", "v_cvt_pkrtz_f16_f32 %1, %4, %5", "v_cvt_pkrtz_f16_f32 %2, %4, %5",      \
      "v_cvt_pkrtz_f16_f32 %3, %4, %5")                                                                                                   \
    X(5, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %6", "v_cvt_f32_f16 %1, %6", "v_cvt_f32_f16 %2, %6", "v_cvt_f32_f16 %3, %6")                  \
    X(6, "v_cvt_f32_f16 sdwa hi", "v_cvt_f32_f16_sdwa %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",                         \
      "v_cvt_f32_f16_sdwa %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",                                                    \
      "v_cvt_f32_f16_sdwa %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",                                                    \
      "v_cvt_f32_f16_sdwa %3, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")                                                    \
    X(7, "v_max_f32", "v_max_f32 %0, %4, %5", "v_max_f32 %1, %4, %5", "v_max_f32 %2, %4, %5", "v_max_f32 %3, %4, %5")                      \
    X(8, "v_max3_f32", "v_max3_f32 %0, %0, %4, %5", "v_max3_f32 %1, %1, %4, %5", "v_max3_f32 %2, %2, %4, %5", "v_max3_f32 %3, %3, %4, %5")   \
    X(9, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %4, 31", "v_alignbit_b32 %1, %1, %4, 31", "v_alignbit_b32 %2, %2, %4, 31",              \
      "v_alignbit_b32 %3, %3, %4, 31")                                                                                                    \
    X(10, "v_pk_mul_f32", "v_pk_mul_f32 %7, %8, %9", "v_pk_mul_f32 %10, %8, %9", "v_pk_mul_f32 %7, %8, %9", "v_pk_mul_f32 %10, %8, %9")    \
    X(11, "v_pk_fma_f32", "v_pk_fma_f32 %7, %8, %9, %7", "v_pk_fma_f32 %10, %8, %9, %10", "v_pk_fma_f32 %7, %8, %9, %7",                  \
      "v_pk_fma_f32 %10, %8, %9, %10")                                                                                                    \
    X(12, "v_pk_add_f32", "v_pk_add_f32 %7, %8, %9", "v_pk_add_f32 %10, %8, %9", "v_pk_add_f32 %7, %8, %9", "v_pk_add_f32 %10, %8, %9")    \
    X(13, "v_pk_max_f16", "v_pk_max_f16 %0, %6, %6", "v_pk_max_f16 %1, %6, %6", "v_pk_max_f16 %2, %6, %6", "v_pk_max_f16 %3, %6, %6")      \
    X(14, "v_pk_mul_f16", "v_pk_mul_f16 %0, %6, %6", "v_pk_mul_f16 %1, %6, %6", "v_pk_mul_f16 %2, %6, %6", "v_pk_mul_f16 %3, %6, %6")      \
    X(15, "v_pk_fma_f16", "v_pk_fma_f16 %0, %6, %6, %0", "v_pk_fma_f16 %1, %6, %6, %1", "v_pk_fma_f16 %2, %6, %6, %2",                    \
      "v_pk_fma_f16 %3, %6, %6, %3")                                                                                                      \
    X(16, "v_perm_b32", "v_perm_b32 %0, %4, %5, %6", "v_perm_b32 %1, %4, %5, %6", "v_perm_b32 %2, %4, %5, %6", "v_perm_b32 %3, %4, %5, %6") \
    X(17, "v_and_or_b32", "v_and_or_b32 %0, %4, %5, %0", "v_and_or_b32 %1, %4, %5, %1", "v_and_or_b32 %2, %4, %5, %2",                    \
      "v_and_or_b32 %3, %4, %5, %3")                                                                                                      \
    X(18, "v_mul_f32", "v_mul_f32 %0, %4, %5", "v_mul_f32 %1, %4, %5", "v_mul_f32 %2, %4, %5", "v_mul_f32 %3, %4, %5")                     \
    X(19, "v_fma_mixhi_f16 (f16 srcC)", "v_fma_mixhi_f16 %0, %4, %5, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]",                                \
      "v_fma_mixhi_f16 %1, %4, %5, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]", "v_fma_mixhi_f16 %2, %4, %5, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]", \
      "v_fma_mixhi_f16 %3, %4, %5, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]")                                                                 \
    X(20, "v_cndmask_b32", "v_cndmask_b32 %0, %4, %5, vcc", "v_cndmask_b32 %1, %4, %5, vcc", "v_cndmask_b32 %2, %4, %5, vcc",             \
      "v_cndmask_b32 %3, %4, %5, vcc")                                                                                                    \
    X(21, "v_cmp_gt_f32 (vcc)", "v_cmp_gt_f32 vcc, %4, %5", "v_cmp_gt_f32 vcc, %5, %4", "v_cmp_gt_f32 vcc, %4, %5", "v_cmp_gt_f32 vcc, %5, %4") \
    X(22, "v_pk_mov_b32", "v_pk_mov_b32 %7, %8, %9", "v_pk_mov_b32 %10, %8, %9", "v_pk_mov_b32 %7, %8, %9", "v_pk_mov_b32 %10, %8, %9")    \
    X(23, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %4", "v_lshl_or_b32 %1, %1, 1, %4", "v_lshl_or_b32 %2, %2, 1, %4",                   \
      "v_lshl_or_b32 %3, %3, 1, %4")                                                                                                      \
    X(24, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 15, %6", "v_pk_ashrrev_i16 %1, 15, %6", "v_pk_ashrrev_i16 %2, 15, %6",                \
      "v_pk_ashrrev_i16 %3, 15, %6")                                                                                                      \
    X(25, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %6, %6", "v_pk_mad_u16 %1, %1, %6, %6", "v_pk_mad_u16 %2, %2, %6, %6",                    \
      "v_pk_mad_u16 %3, %3, %6, %6") \
    X(26, "SEQ bwd mask: v_add_co + v_cndmask (production)", "v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %4, 0, vcc", "v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %2, %5, 0, vcc", "v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %3, %4, 0, vcc", "v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %5, 0, vcc") \
    X(27, "SEQ bwd mask: v_bfe_i32 + v_bfi_b32", "v_bfe_i32 %1, %0, 31, 1\n\tv_bfi_b32 %1, %1, 0, %4", "v_bfe_i32 %2, %0, 30, 1\n\tv_bfi_b32 %2, %2, 0, %5", "v_bfe_i32 %3, %0, 29, 1\n\tv_bfi_b32 %3, %3, 0, %4", "v_bfe_i32 %1, %0, 28, 1\n\tv_bfi_b32 %1, %1, 0, %5") \
    X(28, "SEQ fwd: v_alignbit + v_max (production)", "v_alignbit_b32 %0, %0, %4, 31\n\tv_max_f32 %1, 0, %4", "v_alignbit_b32 %0, %0, %5, 31\n\tv_max_f32 %2, 0, %5", "v_alignbit_b32 %0, %0, %4, 31\n\tv_max_f32 %3, 0, %4", "v_alignbit_b32 %0, %0, %5, 31\n\tv_max_f32 %1, 0, %5") \
    X(29, "SEQ split pair: 4 v_fma_mix (production)", "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %5, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixlo_f16 %1, %4, %5, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]", "v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %2, %5, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixlo_f16 %3, %4, %5, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %5, %5, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]", "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %5, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixlo_f16 %1, %4, %5, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]", "v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %2, %5, %5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixlo_f16 %3, %4, %5, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %5, %5, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]") \
    X(30, "SEQ split pair: 2 mul, cvt_pk, 2 fma_mix_f32, cvt_pk", "v_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %5, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mix_f32 %2, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %1, %2, %3", "v_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %5, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mix_f32 %2, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %1, %2, %3", "v_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %5, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mix_f32 %2, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %1, %2, %3", "v_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %5, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mix_f32 %2, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %1, %2, %3") \
    X(31, "v_bfe_i32", "v_bfe_i32 %0, %4, 31, 1", "v_bfe_i32 %1, %4, 30, 1", "v_bfe_i32 %2, %4, 29, 1", "v_bfe_i32 %3, %4, 28, 1") \
    X(32, "v_bfi_b32", "v_bfi_b32 %0, %4, 0, %5", "v_bfi_b32 %1, %4, 0, %5", "v_bfi_b32 %2, %4, 0, %5", "v_bfi_b32 %3, %4, 0, %5") \
    X(33, "v_add_co_u32 (vcc)", "v_add_co_u32 %0, vcc, %0, %0", "v_add_co_u32 %1, vcc, %1, %1", "v_add_co_u32 %2, vcc, %2, %2", "v_add_co_u32 %3, vcc, %3, %3") \
    X(34, "v_cndmask_b32 (sgpr pair mask, e64)", "v_cndmask_b32_e64 %0, %4, %5, s[20:21]", "v_cndmask_b32_e64 %1, %4, %5, s[20:21]", "v_cndmask_b32_e64 %2, %4, %5, s[20:21]", "v_cndmask_b32_e64 %3, %4, %5, s[20:21]") \
    X(35, "v_accvgpr_read_b32", "v_accvgpr_read_b32 %0, a0", "v_accvgpr_read_b32 %1, a1", "v_accvgpr_read_b32 %2, a2", "v_accvgpr_read_b32 %3, a3")

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void alone(unsigned long long* out, float* sink) {
    float a = threadIdx.x * 0.5f + 1.f, b = 1.25f;
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, h = 0x3c003c00u;
    f32x2 p0 = {a, b}, p1 = {b, a}, pa = {a, a + 1}, pb = {b, b};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 32; ++it) {
#define X(K, NAME, I0, I1, I2, I3)                                                                                                        \
        if (KIND == K) {                                                                                                                  \
            REP32(asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3                                                                           \
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)                                                                   \
                               : "v"(a), "v"(b), "v"(h), "v"(p0), "v"(pa), "v"(pb), "v"(p1)                                               \
                               : "vcc", "s20", "s21", "a0", "a1", "a2", "a3");)                                                                                                 \
        }
        KINDS(X)
#undef X
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = t1 - t0;
    sink[threadIdx.x] = (float)(d0 + d1 + d2 + d3) + p0[0] + p1[1];
}

int main() {
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 4096);
    unsigned long long h[2][64];
    for (int w = 0; w < 2; ++w) {
        const int threads = w ? 512 : 256;
#define X(K, NAME, I0, I1, I2, I3) alone<K><<<256, threads>>>(out, sink);
        KINDS(X)
#undef X
        (void)hipMemcpy(h[w], out, 64 * 8, hipMemcpyDeviceToHost);
    }
#define X(K, NAME, I0, I1, I2, I3)                                                                                                        \
    printf("%-28s %6.2f cycles per instruction with one wave per SIMD, %6.2f per instruction and wave with two\n", NAME,                   \
           (double)h[0][K] / (32.0 * 32 * 4), (double)h[1][K] / (32.0 * 32 * 4));
    KINDS(X)
#undef X
    return 0;
}
