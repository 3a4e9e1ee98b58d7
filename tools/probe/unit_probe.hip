// One 2 KiB weight unit (A_hi | A_lo) against 32 samples, in the two MFMA shapes gfx950 offers for fp16:
//     SHAPE 32: three v_mfma_f32_32x32x16_f16 (A = 32 rows x 16 k, B = 16 k x 32 samples)            -- the field kernels' unit
//     SHAPE 16: six   v_mfma_f32_16x16x32_f16 (A = 16 rows x 32 k, B = 32 k x 16 samples, two sample blocks per A fragment)
//     SHAPE 64: six   v_mfma_f32_32x32x16_f16 against TWO 32-sample blocks (twice the work per weight unit: a Wd = 128 kernel at 64 samples per wave)
// Both do 3 x 16 384 multiply-adds per unit on 1 KiB + 1 KiB of weights read from LDS; everything around the MFMAs is the field
// kernels' (tools/probe/ring_probe.hip): one workgroup of four waves per CU, 32 KiB slabs of sixteen units, two LDS slots refilled
// through registers (global_load_dwordx4 -> ds_write_b128, one piece per odd unit), both A groups read with ds_read_b128 two units
// ahead, lgkmcnt(0) + s_barrier per slab, 128 accumulator registers in AGPRs.  VALU > 0 adds that many independent v_fma_f32 per
// unit, spread over the gaps (the kernels carry about 12: operand split, ReLU, masks, exponents).  Operands are random fp16.
// REFILL: 3 = as the kernels, 1 = the global loads only, 2 = the ds_writes only (stale registers), 0 = neither (slot 0 re-read);
// +4 = no s_barrier per slab, +8 = only the hi group of a unit is read from LDS (half the A-operand traffic).
// The MI355X runs this against its power limit, not its clock limit: the figure of merit is ns per unit.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/unit_probe.hip -o tools/probe/unit_probe && tools/probe/unit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SLAB 32768
#define UNITS 16

template <int N> struct Fma {       // N independent v_fma_f32 on registers the compiler cannot fold
    static __device__ __forceinline__ void run(float (&v)[8], float k) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i % 8]) : "v"(k));
    }
};

template <int SHAPE, int VALU, int REFILL>
__global__ __launch_bounds__(256, 1) void walk(const char* stream, const f32x4* bsrc, int n_slabs, int rounds, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t my_off = (uint32_t)(wave * 8 * 1024 + lane * 16);
    char* my_lds = smem + my_off;
    const char* ring_lane = smem + lane * 16;
    f32x4 stage[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) stage[q] = *(const f32x4*)(stream + my_off + q * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) *(f32x4*)(my_lds + q * 1024) = stage[q];
    uint32_t g_next = n_slabs > 1 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) stage[q] = *(const f32x4*)(stream + (size_t)g_next * SLAB + my_off + q * 1024);
    g_next = g_next + 1 == (uint32_t)n_slabs ? 0 : g_next + 1;
    // B operands: random fp16 of order one; the 16-wide shape needs one (hi, lo) pair per sample block
    f32x4 Bh0 = bsrc[threadIdx.x * 4 + 0], Bl0 = bsrc[threadIdx.x * 4 + 1], Bh1 = bsrc[threadIdx.x * 4 + 2], Bl1 = bsrc[threadIdx.x * 4 + 3];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // 128 accumulator registers either way: 8 tiles of 32 x 32, or 16 row tiles x 2 sample blocks of 16 x 16
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x16 acc2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    f32x4 a16[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) a16[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * lane + i;
    const float kf = 0.999f;
    constexpr int G1 = VALU / 3, G2 = VALU / 3, G3 = VALU - 2 * (VALU / 3);
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
            const int q = uu / 2;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SHAPE == 32) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(l0), "v"(Bh0));
            } else if constexpr (SHAPE == 64) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(l0), "v"(Bh0));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc2[uu % 8]) : "v"(l0), "v"(Bh1));
            } else {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu]) : "v"(l0), "v"(Bh0));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu + 1]) : "v"(l0), "v"(Bh1));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (uu + 2 < UNITS) h2 = *(const f32x4*)(p + (2 * uu + 4) * 1024);
            Fma<G1>::run(v, kf);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(l0), "v"(Bh0), "v"(Bh1));
            if constexpr (SHAPE == 32) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(h0), "v"(Bl0));
            } else if constexpr (SHAPE == 64) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(h0), "v"(Bl0));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc2[uu % 8]) : "v"(h0), "v"(Bl1));
            } else {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu]) : "v"(h0), "v"(Bl0));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu + 1]) : "v"(h0), "v"(Bl1));
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((uu & 1) && (REFILL & 2)) *(f32x4*)(my_lds + idle + q * 1024) = stage[q];
            Fma<G2>::run(v, kf);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(Bl0), "v"(Bl1));
            if constexpr (SHAPE == 32) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(h0), "v"(Bh0));
            } else if constexpr (SHAPE == 64) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[uu % 8]) : "v"(h0), "v"(Bh0));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc2[uu % 8]) : "v"(h0), "v"(Bh1));
            } else {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu]) : "v"(h0), "v"(Bh0));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a16[2 * uu + 1]) : "v"(h0), "v"(Bh1));
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((uu & 1) && (REFILL & 1)) stage[q] = *(const f32x4*)(stream + (size_t)g_next * SLAB + my_off + q * 1024);
            if (uu + 2 < UNITS && !(REFILL & 8)) l2 = *(const f32x4*)(p + (2 * uu + 5) * 1024);
            Fma<G3>::run(v, kf);
            asm volatile("" ::"v"(h0), "v"(Bh0), "v"(Bh1));
            __builtin_amdgcn_sched_barrier(0);
        }
        g_next = g_next + 1 == (uint32_t)n_slabs ? 0 : g_next + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(REFILL & 4)) __builtin_amdgcn_s_barrier();
        if ((REFILL & 3) == 3) c_slot ^= 1u;
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(a16[30]), "+a"(a16[31]), "+a"(a16[28]), "+a"(a16[29]));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc2[0]), "+a"(acc2[1]), "+a"(acc2[2]), "+a"(acc2[3]), "+a"(acc2[4]), "+a"(acc2[5]), "+a"(acc2[6]), "+a"(acc2[7]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = stage[0][0];
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + v[t] + acc2[t][0];
#pragma unroll
    for (int t = 0; t < 32; ++t) s += a16[t][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int VALU, int REFILL = 3>
static double run(const char* stream, const f32x4* bsrc, int n_slabs, int rounds) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, cus * 8); (void)hipMalloc(&sink, cus * 256 * 4);
    auto k = walk<SHAPE, VALU, REFILL>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 2 * SLAB, 0, stream, bsrc, n_slabs, rounds / 8, cyc, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 2 * SLAB, 0, stream, bsrc, n_slabs, rounds, cyc, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = (unsigned long long*)malloc(cus * 8);
    (void)hipMemcpy(h, cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    const double n_units = (double)rounds * UNITS;
    static const char* rf[16] = {"no refill         ", "loads only        ", "ds_writes only    ", "refill            ",
                                 "no refill, no barrier", "", "", "refill, no barrier (racy)", "no refill, half the A reads", "", "", "", "no refill/barrier, half A", "", "", ""};
    printf("%dx%d, %2d VALU per unit, %s(%2d slabs) %7.1f ms  %6.1f cycles/unit  clock %.2f GHz  %6.2f ns/unit  %7.1f TFLOP/s fp16  (%s)\n", SHAPE, SHAPE, VALU, rf[REFILL], n_slabs, ms,
           mean / n_units, mean / ms / 1e6, ms * 1e6 / n_units, n_units * (SHAPE == 64 ? 6 : 3) * 32768 * 4 * cus / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    free(h); (void)hipFree(cyc); (void)hipFree(sink);
    return ms;
}

int main() {
    const int n_slabs = 83;                                   // 2.6 MB, as the headline forward stream
    char* stream; f32x4* bsrc;
    (void)hipMalloc(&stream, (size_t)n_slabs * SLAB);
    (void)hipMalloc(&bsrc, 256 * 4 * 16);
    uint16_t* h = (uint16_t*)malloc((size_t)n_slabs * SLAB);
    srand(3);
    for (size_t i = 0; i < (size_t)n_slabs * SLAB / 2; ++i) h[i] = (uint16_t)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));   // random fp16 of order 1
    (void)hipMemcpy(stream, h, (size_t)n_slabs * SLAB, hipMemcpyHostToDevice);
    for (size_t i = 0; i < 256 * 4 * 8; ++i) h[i] = (uint16_t)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));
    (void)hipMemcpy(bsrc, h, 256 * 4 * 16, hipMemcpyHostToDevice);
    const int rounds = 20000;                                 // x 16 units x ~100 cycles ~ 32 M cycles ~ 20 ms
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0>(stream, bsrc, n_slabs, rounds);
        run<16, 0>(stream, bsrc, n_slabs, rounds);
        run<32, 12>(stream, bsrc, n_slabs, rounds);
        run<16, 12>(stream, bsrc, n_slabs, rounds);
        run<32, 18>(stream, bsrc, n_slabs, rounds);
        run<16, 18>(stream, bsrc, n_slabs, rounds);
        // where the refill's cost sits: its two halves alone, and the same refill from a stream that stays in the CU's own cache
        run<32, 12, 0>(stream, bsrc, n_slabs, rounds);
        run<32, 12, 1>(stream, bsrc, n_slabs, rounds);
        run<32, 12, 2>(stream, bsrc, n_slabs, rounds);
        run<32, 12, 3>(stream, bsrc, 1, rounds);
        // the other parts of the skeleton: the barrier per slab, and the A-operand reads (bit 3: the lo group is not re-read)
        run<32, 12, 4>(stream, bsrc, n_slabs, rounds);
        run<32, 12, 8>(stream, bsrc, n_slabs, rounds);
        run<32, 12, 12>(stream, bsrc, n_slabs, rounds);
        run<32, 0, 12>(stream, bsrc, n_slabs, rounds);
        // "64": the 32x32x16 unit against TWO 32-sample blocks per wave (six MFMAs per A pair: what a Wd = 128 kernel with 64 samples per
        // wave would run; its side work per unit doubles with the samples).  Compare per MFMA with the 18-VALU line above.
        run<64, 24>(stream, bsrc, n_slabs, rounds);
        run<64, 36>(stream, bsrc, n_slabs, rounds);
    }
    return 0;
}
