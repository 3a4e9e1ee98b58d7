// Sustained matrix-core rate of THIS GPU under its power management: every SIMD issues v_mfma_f32_32x32x16_f16 back to back on
// four independent accumulator tiles for a few tens of milliseconds; s_memtime cycles / wall time is the shader clock the chip
// settles on.  With all-zero operands an MI355X holds ~2.35 GHz (the 2.5 PFLOP/s of the data sheet); with operands whose bits
// toggle it falls to ~1.65 GHz = ~1.72 PFLOP/s, and VALU work beside the MFMAs lowers it further (tools/probe/clock_probe.hip:
// the full table, including v_mfma_f32_16x16x32_f16).  bench.py quotes the field kernels against BOTH peaks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nefes_hip.h"

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_clock_kernel(int random, int iters, unsigned long long* cyc, float* sink) {
    f16x8 A, B;
    uint32_t x = 0x9e3779b9u * (threadIdx.x + 1u) + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {            // xorshift: values in (-2, 2), every mantissa bit in play
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        A[i] = random ? (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 16384.f)) : (_Float16)0.f;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        B[i] = random ? (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 16384.f)) : (_Float16)0.f;
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(A), "v"(B));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(A), "v"(B));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(A), "v"(B));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(A), "v"(B));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));      // results readable by the VALU below
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
}  // namespace

// Sixteen-byte stores that re-use their data registers, as compiled code does: a lane stores (v, v, v + j, v) to plane j = 0..7 from the
// SAME four registers, adding 1 to the third one `NOPS` + 1 wait states behind each store (two are what the hardware documents and what
// hipcc leaves; more is always legal).  out[j][i][2] != v + j = that store went out with a LATER value of its register.
// tools/store_hazard.py runs it alone on the device and next to the field kernels of another stream.
template <int NOPS>
__global__ __launch_bounds__(256) void store_hazard_kernel(float* __restrict__ out, long n, int reps) {
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        const long i = i0 + (long)r * gridDim.x * 256;
        if (i >= n) return;
        float* q = out + 4 * i;
        const float v = (float)(i & 0xfffff) + 1.f;
        const long plane = 4 * n * 4;                        // bytes between the eight planes a lane stores to
        asm volatile(
            "v_mov_b32 v10, %1\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %1\n\tv_mov_b32 v13, %1\n\t"
            "v_mov_b32 v14, %0\n\tv_mov_b32 v15, %4\n\t"
            "s_nop 4\n\t"
            ".rept 8\n\t"
            "global_store_dwordx4 v[14:15], v[10:13], off\n\t"
            "s_nop %2\n\t"
            "v_add_f32 v12, 1.0, v12\n\t"
            "v_lshl_add_u64 v[14:15], %3, 0, v[14:15]\n\t"
            ".endr\n\t"
            "s_nop 4"
            :: "v"((uint32_t)(uintptr_t)q), "v"(v), "n"(NOPS), "s"(plane), "v"((uint32_t)((uintptr_t)q >> 32))
            : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
}

extern "C" int nefes_probe_store_hazard(float* out, int64_t n_float4, int nops, void* stream) {
    if (!out || n_float4 <= 0) return NEFES_E_BADARG;
    const int reps = 8;
    const unsigned blocks = (unsigned)((n_float4 + 256 * reps - 1) / (256 * reps));
    hipStream_t st = (hipStream_t)stream;
    switch (nops) {
        case 0: hipLaunchKernelGGL(store_hazard_kernel<0>, dim3(blocks), dim3(256), 0, st, out, (long)n_float4, reps); break;
        case 1: hipLaunchKernelGGL(store_hazard_kernel<1>, dim3(blocks), dim3(256), 0, st, out, (long)n_float4, reps); break;
        case 3: hipLaunchKernelGGL(store_hazard_kernel<3>, dim3(blocks), dim3(256), 0, st, out, (long)n_float4, reps); break;
        case 7: hipLaunchKernelGGL(store_hazard_kernel<7>, dim3(blocks), dim3(256), 0, st, out, (long)n_float4, reps); break;
        case 15: hipLaunchKernelGGL(store_hazard_kernel<15>, dim3(blocks), dim3(256), 0, st, out, (long)n_float4, reps); break;
        default: return NEFES_E_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

// v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1] and [1,0] (the low result takes the HIGH half of one operand): the form that produced
// composite_bwd4_kernel's wrong elements (section 4.7 of DESIGN.md).  Every lane runs `iters` of them on fresh operands and counts the
// results that are not the two products (bit-exact); out[i] = four byte-wide counts (iters <= 255).  tools/store_hazard.py runs it next to another stream's
// field kernels.
__global__ __launch_bounds__(256) void pk_mul_probe_kernel(unsigned* __restrict__ out, long n, int iters) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned bad = 0;
    float a = (float)(i & 1023) * 0.37f + 1.f, b = (float)(i & 511) * 0.11f + 2.f;
    for (int k = 0; k < iters; ++k) {
        const float2 x = make_float2(a, b), y = make_float2(3.f + (float)(k & 7), 5.f + (float)(k & 3));
        float2 r01, r10, s01, s10;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r01) : "v"(x), "v"(y));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r10) : "v"(x), "v"(y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(s01) : "v"(x), "v"(y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(s10) : "v"(x), "v"(y));
        // op_sel = which half feeds the LOW result (op_sel_hi, default [1,1], feeds the high one).  One byte per instruction: wrong results
        // in either half -- mul [0,1], mul [1,0], add [0,1], add [1,0] (saturating at 255)
        const unsigned e0 = (r01.x != a * y.y) | (r01.y != b * y.y), e1 = (r10.x != b * y.x) | (r10.y != b * y.y);
        const unsigned e2 = (s01.x != a + y.y) | (s01.y != b + y.y), e3 = (s10.x != b + y.x) | (s10.y != b + y.y);
        bad += e0 + (e1 << 8) + (e2 << 16) + (e3 << 24);
        a += 0.25f; b += 0.5f;
    }
    out[i] = bad;
}

// A neighbour for pk_mul_probe_kernel: `iters` rounds of ONE instruction kind on every SIMD of the device, to find which instruction
// of the field kernels it is that another wave's v_pk_mul_f32 op_sel:[0,1] does not survive.
//   0: v_fma_mixlo_f16 / v_fma_mixhi_f16 (the fp16 two-part split's conversions: VOP3P with op_sel, like the victim)
//   1: v_mfma_f32_32x32x16_f16 back to back      2: both, alternating      3: v_pk_fma_f32 with op_sel_hi:[1,0,1]
//   4: plain v_fma_f32 (control)
template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(float* __restrict__ sink, int iters) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    float x = (float)threadIdx.x * 0.001f + 1.f, y = 0.5f, z = 0.25f;
    unsigned h = 0;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.01f * (float)(threadIdx.x + r)); b[r] = (_Float16)(0.02f * (float)r); }
    float2 p = make_float2(x, y), q = make_float2(z, x);
    asm volatile("s_nop 1" : "+v"(acc), "+v"(a), "+v"(b));       // (vector writes of the operands -> the first asm MFMA: two wait states)
    for (int k = 0; k < iters; ++k) {
        if (KIND == 0 || KIND == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %2, %1, 0" : "+v"(h) : "v"(x), "v"(y));
        }
        if (KIND == 1 || KIND == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
        if (KIND == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(q), "v"(q));
        }
        if (KIND == 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z) : "v"(x), "v"(y));
        }
    }
    asm volatile("s_nop 11" : "+v"(acc));                         // (the last asm MFMA -> vector reads of its result: twelve)
    if (sink) sink[(long)blockIdx.x * 256 + threadIdx.x] = (float)h + acc[0] + acc[7] + p.x + p.y + z;
}

extern "C" int nefes_probe_aggressor(int kind, int iters, int blocks, float* sink, void* stream) {
    if (iters <= 0 || blocks <= 0 || !sink) return NEFES_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        case 0: hipLaunchKernelGGL(aggressor_kernel<0>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
        case 1: hipLaunchKernelGGL(aggressor_kernel<1>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
        case 2: hipLaunchKernelGGL(aggressor_kernel<2>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
        case 3: hipLaunchKernelGGL(aggressor_kernel<3>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
        case 4: hipLaunchKernelGGL(aggressor_kernel<4>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
        default: return NEFES_E_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

extern "C" int nefes_probe_pk_mul(unsigned* out, int64_t n, int iters, void* stream) {
    if (!out || n <= 0 || iters <= 0) return NEFES_E_BADARG;
    hipLaunchKernelGGL(pk_mul_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long)n, iters);
    return (int)hipGetLastError();
}

extern "C" int nefes_probe_mfma_clock(int random_operands, int ms_target, double* clock_ghz, double* fp16_dense_tflops,
                                      void* stream) {
    if (!clock_ghz || !fp16_dense_tflops || ms_target <= 0 || ms_target > 2000) return NEFES_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    unsigned long long* cyc = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&cyc, (size_t)cus * 8) != hipSuccess || hipMalloc(&sink, (size_t)cus * 256 * 4) != hipSuccess) return NEFES_E_BADARG;
    const int iters = ms_target * 2000;          // 32 MFMAs x 32 cycles per iteration ~ 0.5 us at 2 GHz
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_clock_kernel, dim3(cus), dim3(256), 0, st, random_operands, iters / 4 + 1, cyc, sink);   // ramp
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_clock_kernel, dim3(cus), dim3(256), 0, st, random_operands, iters, cyc, sink);
    (void)hipEventRecord(e1, st);
    int rc = (int)hipGetLastError();
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = new unsigned long long[cus];
    (void)hipMemcpy(h, cyc, (size_t)cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    delete[] h;
    (void)hipFree(cyc);
    (void)hipFree(sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != 0 || ms <= 0.f) return rc ? rc : NEFES_E_BADARG;
    *clock_ghz = mean / ms / 1e6;
    *fp16_dense_tflops = (double)iters * 32.0 * (2.0 * 32 * 32 * 16) * 4.0 * cus / ms / 1e9;
    return 0;
}
