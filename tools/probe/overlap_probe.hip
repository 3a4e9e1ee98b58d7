// Micro-probe (gfx950): does VALU work of a SECOND wave on the same SIMD overlap with fp32 / bf16 MFMAs of the first?
// 8 waves per workgroup (waves w and w+4 share SIMD w%4): waves 0-3 run an MFMA-only loop, waves 4-7 a VALU-only loop.
// Reports cycles for: MFMA alone, VALU alone, both.  both ~ max => overlap, both ~ sum => serial.
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(512, 1) void probe(float* out, unsigned long long* cyc, int n_mfma, int n_valu, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = out[threadIdx.x], b = out[threadIdx.x + 512];
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)a; bh[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma) {
        for (int it = 0; it < n_mfma / 4; ++it) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int it = 0; it < n_valu / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += acc[t][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 16 + wave * 2] = t1 - t0; cyc[blockIdx.x * 16 + wave * 2 + 1] = t2 - t0; }
}

// Same wave: one MFMA followed by NV independent VALU instructions, 4 waves per workgroup (one per SIMD).
template <int KIND, int NV>
__global__ __launch_bounds__(256, 1) void interleave(float* out, unsigned long long* cyc, int n_mfma) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = out[threadIdx.x], b = out[threadIdx.x + 512];
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)a; bh[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n_mfma / 4; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += acc[t][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV>
static void run_il(const char* name, float* d_out, unsigned long long* d_cyc) {
    unsigned long long h;
    const int n = 4096;
    hipLaunchKernelGGL((interleave<KIND, NV>), dim3(256), dim3(256), 0, 0, d_out, d_cyc, n);
    hipLaunchKernelGGL((interleave<KIND, NV>), dim3(256), dim3(256), 0, 0, d_out, d_cyc, n);
    hipDeviceSynchronize();
    hipMemcpy(&h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-14s same wave, %d VALU per MFMA: %.1f cycles per MFMA\n", name, NV, (double)h / n);
}

// bf16x6 feasibility: per "slab" 96 bf16 MFMAs (16 units x 6), 48 ds_read_b128 (3 per unit), P LDS-DMA pieces per wave
// (1 KiB each, from an L2-resident blob), one counted wait + workgroup barrier.  4 waves, one per SIMD.
// DV = 0: padded (s_nop 4) + M0 saved/restored; 1: scalar copy of the base instead of the padding (what the kernels do);
// 2: as 1 without restoring M0 (nothing else in these kernels reads it)
template <int DV>
__device__ __forceinline__ void dma16(const void* gbase, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    unsigned long long base2;
    if (DV == 0)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else if (DV == 1)
        asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(base2) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b64 %0, %2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                     : "=&s"(base2) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
}
template <int P, int NV, int IL = 0, int DV = 0>
__global__ __launch_bounds__(256, 1) void dma_mix(const char* blob, float* out, unsigned long long* cyc, int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = out[threadIdx.x] + i;
    bf16x8 bh;
    for (int i = 0; i < 8; ++i) bh[i] = (__bf16)v[i];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane_off = wave * 1024 + lane * 16;
    const char* rd = smem + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int sl = 0; sl < n_slabs; ++sl) {
        const int slot = sl % 3;
        const char* src = blob + (size_t)(sl & 63) * 49152;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x4 fa = *(const f32x4*)(rd + slot * 49152 + (3 * u) * 1024);
            f32x4 fb = *(const f32x4*)(rd + slot * 49152 + (3 * u + 1) * 1024);
            f32x4 fc = *(const f32x4*)(rd + slot * 49152 + (3 * u + 2) * 1024);
            bf16x8 a0, a1, a2;
            __builtin_memcpy(&a0, &fa, 16); __builtin_memcpy(&a1, &fb, 16); __builtin_memcpy(&a2, &fc, 16);
            const int x0 = u & 7, x1 = IL ? ((u + 4) & 7) : (u & 7);     // IL: alternate between two accumulators
            acc[x0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[x0], 0, 0, 0);
            if (u * P / 16 != (u + 1) * P / 16) {
                __builtin_amdgcn_sched_barrier(0);
                for (int q = u * P / 16; q < (u + 1) * P / 16; ++q)
                    dma16<DV>(src + q * 4096, lane_off, lds0 + ((slot + 2) % 3) * 49152 + wave * 1024 + q * 4096);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[x1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[x1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
            __builtin_amdgcn_sched_barrier(0);
            acc[x0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[x1], 0, 0, 0);
            acc[x0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bh, acc[x1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(P) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int P, int NV, int IL = 0, int DV = 0>
static void run_mix(float* d_out, unsigned long long* d_cyc, const char* blob) {
    unsigned long long h;
    const int n = 200;
    hipFuncSetAttribute((const void*)dma_mix<P, NV, IL, DV>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((dma_mix<P, NV, IL, DV>), dim3(256), dim3(256), 3 * 49152, 0, blob, d_out, d_cyc, n);
    hipDeviceSynchronize();
    hipMemcpy(&h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s[dma variant %d] ", IL ? "[two accumulators alternating] " : "", DV); printf("bf16x6 mix: %2d DMA pieces + 48 ds_read_b128 + %2d VALU/unit per 96 MFMAs: %.1f cycles per MFMA (floor 32)\n", P, NV, (double)h / (n * 96.0));
}

template <int KIND>
static void run(const char* name, float* d_out, unsigned long long* d_cyc, int n_mfma, int n_valu) {
    unsigned long long h[16];
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d_out, d_cyc, n_mfma, n_valu, mode);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d_out, d_cyc, n_mfma, n_valu, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s mode %d (%s): wave0 %llu cyc, wave4 %llu cyc, workgroup %llu cyc   [%d MFMA, %d VALU per wave]\n", name, mode,
               mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", h[0], h[8], h[1] > h[9] ? h[1] : h[9], n_mfma, n_valu);
    }
}

int main() {
    float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, 4096 * sizeof(float)); hipMemset(d_out, 0, 4096 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 16 * sizeof(unsigned long long));
    run<0>("fp32 32x32x2 + v_max", d_out, d_cyc, 4096, 16384);      // 4096 x 64 = 262k MFMA cycles; 16384 VALU
    run<0>("fp32 32x32x2 + v_max (x4)", d_out, d_cyc, 4096, 65536);
    run<1>("bf16 32x32x16 + v_max", d_out, d_cyc, 8192, 16384);
    run<1>("bf16 32x32x16 + v_max (x4)", d_out, d_cyc, 8192, 65536);
    run_il<0, 0>("fp32 32x32x2", d_out, d_cyc); run_il<0, 1>("fp32 32x32x2", d_out, d_cyc); run_il<0, 2>("fp32 32x32x2", d_out, d_cyc);
    run_il<0, 4>("fp32 32x32x2", d_out, d_cyc); run_il<0, 8>("fp32 32x32x2", d_out, d_cyc);
    run_il<1, 0>("bf16 32x32x16", d_out, d_cyc); run_il<1, 1>("bf16 32x32x16", d_out, d_cyc); run_il<1, 2>("bf16 32x32x16", d_out, d_cyc);
    run_il<1, 4>("bf16 32x32x16", d_out, d_cyc); run_il<1, 6>("bf16 32x32x16", d_out, d_cyc); run_il<1, 8>("bf16 32x32x16", d_out, d_cyc);
    char* blob; hipMalloc(&blob, 64 * 49152); hipMemset(blob, 0, 64 * 49152);
    run_mix<0, 0>(d_out, d_cyc, blob); run_mix<12, 0>(d_out, d_cyc, blob); run_mix<12, 8>(d_out, d_cyc, blob); run_mix<12, 16>(d_out, d_cyc, blob);
    run_mix<6, 8>(d_out, d_cyc, blob);
    run_mix<0, 0, 1>(d_out, d_cyc, blob); run_mix<12, 8, 1>(d_out, d_cyc, blob);
    run_mix<12, 0, 0, 1>(d_out, d_cyc, blob); run_mix<12, 8, 0, 1>(d_out, d_cyc, blob); run_mix<12, 0, 0, 2>(d_out, d_cyc, blob); run_mix<12, 8, 0, 2>(d_out, d_cyc, blob);
    run_mix<12, 8, 1, 2>(d_out, d_cyc, blob); run_mix<24, 8, 0, 2>(d_out, d_cyc, blob);
    return 0;
}
