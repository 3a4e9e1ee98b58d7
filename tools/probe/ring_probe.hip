// Weight-ring refill under the asm MFMA schedule: one workgroup of four waves per CU walks a packed weight
// stream exactly as the fp16 field kernels do -- 32 KiB slabs of sixteen 2 KiB units (A_hi | A_lo), two LDS slots, every unit three
// `asm volatile` v_mfma_f32_32x32x16_f16 on eight rotating AGPR tiles, both A groups of a unit read with ds_read_b128 two units
// ahead, lgkmcnt(0) + s_barrier per slab -- and refills the idle slot in one of three ways:
//     0  no refill (the floor)
//     1  through registers, as production (field_h3.h StagedRing): per wave and slab eight global_load_dwordx4 and, one slab later,
//        eight ds_write_b128, one of each per odd unit
//     2  LDS-DMA: eight global_load_lds_dwordx4 per wave and slab into the idle slot, one per odd unit, vmcnt(0) before the barrier
// Reports s_memtime cycles per MFMA, the clock and the time per MFMA.  The stream is 2.6 MB and stays in L2, as in the kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/ring_probe.hip -o tools/probe/ring_probe && tools/probe/ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SLAB 32768
#define UNITS 16

__device__ __forceinline__ void dma16(const void* gbase, uint32_t lane_off, uint32_t lds_dst) {
    uint64_t base2;
    asm volatile("s_mov_b64 %0, %2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 : "=&s"(base2) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void walk(const char* stream, int n_slabs, int rounds, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t my_off = (uint32_t)(wave * 8 * 1024 + lane * 16);
    char* my_lds = smem + my_off;
    const char* ring_lane = smem + lane * 16;
    f32x4 stage[8];
    // slab 0 -> slot 0 (through registers in every mode), slab 1 requested
#pragma unroll
    for (int q = 0; q < 8; ++q) stage[q] = *(const f32x4*)(stream + my_off + q * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) *(f32x4*)(my_lds + q * 1024) = stage[q];
    uint32_t g_next = n_slabs > 1 ? 1 : 0;
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage[q] = *(const f32x4*)(stream + (size_t)g_next * SLAB + my_off + q * 1024);
        g_next = g_next + 1 == (uint32_t)n_slabs ? 0 : g_next + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 Bh, Bl;
    for (int i = 0; i < 4; ++i) { Bh[i] = __uint_as_float(0x3c003c00u + lane * 0x00010001u); Bl[i] = __uint_as_float(0x14001400u + lane * 0x00030001u); }
    uint32_t c_slot = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int rd = 0; rd < rounds; ++rd) {
        const char* p = ring_lane + c_slot * SLAB;
        const uint32_t idle = (c_slot ^ 1u) * SLAB;
        f32x4 ha0 = *(const f32x4*)p, la0 = *(const f32x4*)(p + 1024), ha1 = *(const f32x4*)(p + 2048), la1 = *(const f32x4*)(p + 3072), ha2, la2;
#pragma unroll
        for (int uu = 0; uu < UNITS; ++uu) {
            f32x4 &h0 = uu % 3 == 0 ? ha0 : (uu % 3 == 1 ? ha1 : ha2), &l0 = uu % 3 == 0 ? la0 : (uu % 3 == 1 ? la1 : la2);
            f32x4 &h2 = (uu + 2) % 3 == 0 ? ha0 : ((uu + 2) % 3 == 1 ? ha1 : ha2), &l2 = (uu + 2) % 3 == 0 ? la0 : ((uu + 2) % 3 == 1 ? la1 : la2);
            const int t = uu % 8, q = uu / 2;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(l0), "v"(Bh));
            __builtin_amdgcn_sched_barrier(0);
            if (uu + 2 < UNITS) h2 = *(const f32x4*)(p + (2 * uu + 4) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(l0), "v"(Bh));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(h0), "v"(Bl));
            __builtin_amdgcn_sched_barrier(0);
            if ((uu & 1) && MODE == 1) *(f32x4*)(my_lds + idle + q * 1024) = stage[q];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(Bl));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(h0), "v"(Bh));
            __builtin_amdgcn_sched_barrier(0);
            if ((uu & 1) && MODE == 1) stage[q] = *(const f32x4*)(stream + (size_t)g_next * SLAB + my_off + q * 1024);
            if ((uu & 1) && MODE == 2) dma16(stream + (size_t)g_next * SLAB + q * 1024, my_off, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + idle + wave * 8192 + q * 1024));
            if (uu + 2 < UNITS) l2 = *(const f32x4*)(p + (2 * uu + 5) * 1024);
            asm volatile("" ::"v"(h0), "v"(Bh));
            __builtin_amdgcn_sched_barrier(0);
        }
        g_next = g_next + 1 == (uint32_t)n_slabs ? 0 : g_next + 1;
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (MODE != 0) c_slot ^= 1u;
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    if (MODE == 1) s += stage[0][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* what, const char* stream, int n_slabs, int rounds) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, cus * 8); (void)hipMalloc(&sink, cus * 256 * 4);
    auto k = walk<MODE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 2 * SLAB, 0, stream, n_slabs, rounds / 8, cyc, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 2 * SLAB, 0, stream, n_slabs, rounds, cyc, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = (unsigned long long*)malloc(cus * 8);
    (void)hipMemcpy(h, cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    const double n_mfma = (double)rounds * UNITS * 3;
    printf("%-46s %7.1f ms  %5.1f cycles/MFMA  clock %.2f GHz  %5.2f ns/MFMA  (%s)\n", what, ms, mean / n_mfma, mean / ms / 1e6, ms * 1e6 / n_mfma,
           hipGetErrorString(hipGetLastError()));
    free(h); (void)hipFree(cyc); (void)hipFree(sink);
}

int main() {
    const int n_slabs = 83;                                   // 2.6 MB, as the headline forward stream
    char* stream;
    (void)hipMalloc(&stream, (size_t)n_slabs * SLAB);
    uint16_t* h = (uint16_t*)malloc((size_t)n_slabs * SLAB);
    srand(3);
    for (size_t i = 0; i < (size_t)n_slabs * SLAB / 2; ++i) h[i] = (uint16_t)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));   // random fp16 of order 1
    (void)hipMemcpy(stream, h, (size_t)n_slabs * SLAB, hipMemcpyHostToDevice);
    const int rounds = 20000;                                 // x 48 MFMAs x ~40 cycles ~ 38 M cycles ~ 20 ms
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("no refill (floor)", stream, n_slabs, rounds);
        run<1>("through registers: load + ds_write_b128", stream, n_slabs, rounds);
        run<2>("LDS-DMA: global_load_lds_dwordx4", stream, n_slabs, rounds);
    }
    return 0;
}
