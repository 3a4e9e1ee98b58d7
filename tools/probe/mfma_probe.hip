// Micro-probe (gfx950): cycles per v_mfma_f32_32x32x2_f32 for the instruction mixes used by the field kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float prev[16];
    for (int r = 0; r < 16; ++r) prev[r] = out[r * 256 + threadIdx.x];
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)smem)[i] = (float)(i & 7);
    __syncthreads();
    const char* p = smem + lane * 16;
    float b = prev[0], bnext = prev[1];
    f32x16 pacc[2];
    for (int t = 0; t < 2; ++t) { for (int r = 0; r < 16; ++r) pacc[t][r] = 0.f; pacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(prev[t], prev[t + 1], pacc[t], 0, 0, 0); }
    unsigned bits = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 a = *(const f32x4*)(p);
#pragma unroll
        for (int s = 0; s < 8; ++s) {            // 8 k-steps x 8 tiles = 64 MFMAs = one "slab"
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x4 an;
                if (VARIANT >= 1) an = *(const f32x4*)(p + ((s * 2 + g + 1) & 15) * 1024); else an = a;
                if (VARIANT == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0 && VARIANT == 3) {    // consumer-side ReLU from a stable VGPR array: 2 VALU ops per k-step
                    b = fmaxf(prev[(s * 2 + it) & 15], 0.f);
                }
                if (g == 0 && VARIANT == 5) {    // + mask capture: ~5 VALU ops per k-step
                    const float v = prev[(s * 2 + it) & 15];
                    bits |= (v > 0.f ? 1u : 0u) << s;
                    b = fmaxf(v, 0.f);
                }
                if (g == 0 && VARIANT == 6) {    // ReLU + mask from a stable ACCUMULATOR-file array (v_accvgpr_read path)
                    const float v = pacc[s & 1][(s * 2 + it) & 15];
                    bits |= (v > 0.f ? 1u : 0u) << s;
                    b = fmaxf(v, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[g * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b, acc[g * 4 + q], 0, 0, 0);
                    if (VARIANT == 8 && g == 0 && q == 1) {   // next k-step's operand, produced 6 MFMAs ahead of its use
                        __builtin_amdgcn_sched_barrier(0);
                        const float v = pacc[s & 1][(s * 2 + it) & 15];
                        bits |= (v > 0.f ? 1u : 0u) << s;
                        bnext = fmaxf(v, 0.f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (VARIANT == 9 && g == 0 && q == 1) {   // same, ReLU only (2 VALU)
                        __builtin_amdgcn_sched_barrier(0);
                        bnext = fmaxf(pacc[s & 1][(s * 2 + it) & 15], 0.f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if ((VARIANT == 8 || VARIANT == 9) && g == 1) b = bnext;
                a = an;
            }
        }
        if (VARIANT == 4) __builtin_amdgcn_s_barrier();
        if (VARIANT == 7 && (it & 15) == 15) {   // serial epilogue: 128 x (read, max, cmp, cndmask, or) every 1024 MFMAs
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[t][r]; bits |= (v > 0.f ? 1u : 0u) << r; acc[t][r] = fmaxf(v, 0.f); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)bits;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    s += pacc[0][3] + pacc[1][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, float* out, unsigned long long* cyc, int iters) {
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 65536, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    m /= 256;
    // s_memtime ticks at a fixed 100 MHz; report wall per MFMA in ns and the implied cycles at 2.4 GHz
    printf("%-44s memtime ticks/MFMA %.3f  => %.1f cycles @2.4GHz\n", name, m / ((double)iters * 64), m / ((double)iters * 64) * 24.0);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4 * 16); hipMalloc(&cyc, 256 * 8);
    hipMemset(out, 0, 256 * 256 * 4 * 16);
    const int iters = 20000;
    run<0>("0: MFMA only", out, cyc, iters);
    run<1>("1: + ds_read_b128 prefetch, compiler waitcnt", out, cyc, iters);
    run<2>("2: + forced lgkmcnt(0) after each ds_read", out, cyc, iters);
    run<3>("3: + lazy ReLU (2 VALU/k-step) from VGPRs", out, cyc, iters);
    run<5>("5: + lazy ReLU + mask (5 VALU/k-step) from VGPRs", out, cyc, iters);
    run<6>("6: + lazy ReLU + mask from accumulator regs", out, cyc, iters);
    run<8>("8: lazy ReLU+mask one k-step AHEAD, behind MFMA #2", out, cyc, iters);
    run<9>("9: lazy ReLU only one k-step AHEAD", out, cyc, iters);
    run<4>("4: + s_barrier per 64 MFMAs", out, cyc, iters);
    run<7>("7: serial epilogue (128x5 VALU) per 1024 MFMAs", out, cyc, iters);
    return 0;
}
