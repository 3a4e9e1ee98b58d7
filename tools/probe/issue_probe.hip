// Issue cost of the instructions the fp16 two-part split is built from, one wave per SIMD (256 threads per CU), measured as
// s_memtime cycles per instruction over an unrolled stream of independent instructions -- alone, and between dependent
// v_mfma_f32_32x32x16_f16 (one MFMA per K instructions).  hipcc --offload-arch=gfx950 -O2 tools/probe/issue_probe.hip -o /tmp/ip && /tmp/ip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void alone(unsigned long long* out, float* sink) {
    float a = threadIdx.x * 0.5f + 1.f, b = 1.25f;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 16; ++it) {
        if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f1) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f2) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f3) : "v"(a), "v"(b));) }
        if (KIND == 1) { REP64(asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(a), "v"(b)); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h1) : "v"(a), "v"(b)); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h2) : "v"(a), "v"(b)); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 2) { REP64(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(a), "v"(b)); asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(a), "v"(b)); asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h2) : "v"(a), "v"(b)); asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 3) { REP64(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(a), "v"(b)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(a), "v"(b)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(a), "v"(b)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(a), "v"(b));) }
        if (KIND == 4) { REP64(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(h0) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(h1) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(h2) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(h3) : "v"(a));) }
        if (KIND == 5) { REP64(asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h0) : "v"(a)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h1) : "v"(a)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h2) : "v"(a)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h3) : "v"(a));) }
        if (KIND == 6) { REP64(asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h0) : "v"(h1), "v"(h2)); asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h3) : "v"(h1), "v"(h2)); asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h0) : "v"(h1), "v"(h2)); asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h3) : "v"(h1), "v"(h2));) }
        if (KIND == 7) { REP64(asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h0) : "v"(h1), "v"(h2)); asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h3) : "v"(h1), "v"(h2)); asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h0) : "v"(h1), "v"(h2)); asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h3) : "v"(h1), "v"(h2));) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = t1 - t0;
    sink[threadIdx.x] = f0 + f1 + f2 + f3 + (float)(h0 + h1 + h2 + h3);
}

// K filler instructions of KIND after each of a chain of dependent MFMAs
template <int KIND, int K>
__global__ __launch_bounds__(256) void between(unsigned long long* out, float* sink) {
    float a = threadIdx.x * 0.5f + 1.f, b = 1.25f;
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float f[8] = {a, a, a, a, a, a, a, a};
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.01f * i); B[i] = (_Float16)(0.02f * i); }
    f32x16 c = {0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k % 8]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h[k % 8]) : "v"(a), "v"(b));
                if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k % 8]) : "v"(a), "v"(b));
                if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[k % 8]) : "v"(a), "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)h[i];
    sink[threadIdx.x] = s + c[0];
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096);
    unsigned long long h[16];
    const char* names[8] = {"v_fma_f32", "v_fma_mixlo_f16", "v_cvt_pkrtz_f16_f32", "v_max3_f32", "v_alignbit_b32", "v_cvt_f16_f32", "v_pk_mul_f16", "v_pk_fma_f16"};
    alone<0><<<256, 256>>>(out, sink); alone<1><<<256, 256>>>(out, sink); alone<2><<<256, 256>>>(out, sink); alone<3><<<256, 256>>>(out, sink);
    alone<4><<<256, 256>>>(out, sink); alone<5><<<256, 256>>>(out, sink); alone<6><<<256, 256>>>(out, sink); alone<7><<<256, 256>>>(out, sink);
    hipMemcpy(h, out, 8 * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < 8; ++k) printf("alone   %-22s %.2f cycles per instruction\n", names[k], (double)h[k] / (16.0 * 64 * 4));
#define RUN(KIND, K) between<KIND, K><<<256, 256>>>(out, sink); hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); printf("between %-22s K=%d per MFMA gap: %.1f cycles per MFMA\n", names[KIND], K, (double)h[0] / (64.0 * 16));
    RUN(0, 0) RUN(0, 2) RUN(0, 4) RUN(0, 6) RUN(0, 8)
    RUN(1, 2) RUN(1, 4) RUN(1, 6) RUN(1, 8)
    RUN(2, 2) RUN(2, 4) RUN(2, 6)
    RUN(3, 4) RUN(3, 6)
    return 0;
}
