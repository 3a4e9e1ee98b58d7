// Sustained shader clock under matrix-core load: every SIMD of the chip runs v_mfma_f32_32x32x16_f16 back to back (four independent
// accumulator tiles per wave) for tens of milliseconds; s_memtime cycles / wall time = the clock the power management settles on.
// Run with zero and with random operands (the power an MFMA draws depends on the bits that toggle), and with a VALU/LDS mix.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/clock_probe.hip -o tools/probe/clock_probe && tools/probe/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int GAP>   // GAP: v_fma_f32 per MFMA (side work)
__global__ __launch_bounds__(256) void burn(const _Float16* in, unsigned long long* cyc, float* sink, int iters) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = in[(threadIdx.x * 8 + i) & 4095]; B[i] = in[(threadIdx.x * 8 + i + 2048) & 4095]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float f[4] = {1.f, 2.f, 3.f, 4.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(A), "v"(B));
            for (int g = 0; g < GAP; ++g) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[g & 3]) : "v"(f[(g + 1) & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(A), "v"(B));
            for (int g = 0; g < GAP; ++g) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[g & 3]) : "v"(f[(g + 1) & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(A), "v"(B));
            for (int g = 0; g < GAP; ++g) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[g & 3]) : "v"(f[(g + 1) & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(A), "v"(B));
            for (int g = 0; g < GAP; ++g) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[g & 3]) : "v"(f[(g + 1) & 3]));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f[0] + f[1] + f[2] + f[3];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same FLOPs per wave with v_mfma_f32_16x16x32_f16 (16 cycles each, eight independent 16x16 tiles)
__global__ __launch_bounds__(256) void burn16(const _Float16* in, unsigned long long* cyc, float* sink, int iters) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = in[(threadIdx.x * 8 + i) & 4095]; B[i] = in[(threadIdx.x * 8 + i + 2048) & 4095]; }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[k]) : "v"(A), "v"(B));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}


// operands that CHANGE from MFMA to MFMA (four A/B register pairs in rotation), as in a real kernel
__global__ __launch_bounds__(256) void burnR(const _Float16* in, unsigned long long* cyc, float* sink, int iters) {
    f16x8 A[4], B[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) { A[k][i] = in[(threadIdx.x * 8 + i + 512 * k) & 4095]; B[k][i] = in[(threadIdx.x * 8 + i + 2048 + 384 * k) & 4095]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(A[0]), "v"(B[m & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(A[1]), "v"(B[(m + 1) & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(A[2]), "v"(B[(m + 2) & 3]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(A[3]), "v"(B[(m + 3) & 3]));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void burn16R(const _Float16* in, unsigned long long* cyc, float* sink, int iters) {
    f16x8 A[4], B[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) { A[k][i] = in[(threadIdx.x * 8 + i + 512 * k) & 4095]; B[k][i] = in[(threadIdx.x * 8 + i + 2048 + 384 * k) & 4095]; }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[k]) : "v"(A[k & 3]), "v"(B[(k + m) & 3]));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// The fp16 two-part unit of field_h3.h: three MFMAs per (tile, k-step) on operands (Al, Ah) x (Bh, Bl); 8 tiles share a B pair, the A
// pair changes from tile to tile, a new B pair every 8 tiles.  ORDER 0 = production (Al Bh, Ah Bl, Ah Bh: both operands change between
// the first two); 1 = (Al Bh, Ah Bh, Ah Bl: one operand held at every step inside a unit, both change into the next unit);
// 2 = order 1 with odd tiles mirrored (Al' Bl, Ah' Bl, Ah' Bh): one operand held at EVERY step.
template <int ORDER>
__global__ __launch_bounds__(256) void burnU(const _Float16* in, unsigned long long* cyc, float* sink, int iters) {
    f16x8 Ah[4], Al[4], Bh[2], Bl[2];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            Ah[k][i] = in[(threadIdx.x * 8 + i + 512 * k) & 4095];
            Al[k][i] = in[(threadIdx.x * 8 + i + 512 * k + 1777) & 4095] * (_Float16)0.001f;
        }
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 8; ++i) {
            Bh[k][i] = in[(threadIdx.x * 8 + i + 2048 + 384 * k) & 4095];
            Bl[k][i] = in[(threadIdx.x * 8 + i + 3001 + 384 * k) & 4095] * (_Float16)0.001f;
        }
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) c[t] = f32x16{0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define MF(C_, A_, B_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C_) : "v"(A_), "v"(B_))
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {            // a new B pair
#pragma unroll
            for (int t = 0; t < 4; ++t) {        // tiles (32 MFMAs per iteration would be 2 x 4 x 3 = 24: iters scaled by the caller)
                if (ORDER == 0) { MF(c[t], Al[t], Bh[q]); MF(c[t], Ah[t], Bl[q]); MF(c[t], Ah[t], Bh[q]); }
                else if (ORDER == 1 || (t & 1) == 0) { MF(c[t], Al[t], Bh[q]); MF(c[t], Ah[t], Bh[q]); MF(c[t], Ah[t], Bl[q]); }
                else { MF(c[t], Al[t], Bl[q]); MF(c[t], Ah[t], Bl[q]); MF(c[t], Ah[t], Bh[q]); }
            }
        }
    }
#undef MF
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int ORDER>
static void runU(const char* what, const _Float16* in, int iters) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, cus * 8); (void)hipMalloc(&sink, cus * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    burnU<ORDER><<<cus, 256>>>(in, cyc, sink, iters / 8);
    (void)hipEventRecord(e0);
    burnU<ORDER><<<cus, 256>>>(in, cyc, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = (unsigned long long*)malloc(cus * 8);
    (void)hipMemcpy(h, cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    const double n_mfma = (double)iters * 24;
    printf("%-34s %6.1f ms  %5.1f cycles/MFMA  clock %.2f GHz  %7.1f TFLOP/s fp16 dense\n", what, ms, mean / n_mfma, mean / ms / 1e6,
           n_mfma * 2.0 * 32 * 32 * 16 * 4 * cus / ms / 1e9);
    free(h); (void)hipFree(cyc); (void)hipFree(sink);
}

template <int GAP>
static void run(const char* what, const _Float16* in, int iters) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, cus * 8); (void)hipMalloc(&sink, cus * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (GAP == -2) burnR<<<cus, 256>>>(in, cyc, sink, iters / 8); else if (GAP == -3) burn16R<<<cus, 256>>>(in, cyc, sink, iters / 8); else if (GAP < 0) burn16<<<cus, 256>>>(in, cyc, sink, iters / 8); else burn<(GAP < 0 ? 0 : GAP)><<<cus, 256>>>(in, cyc, sink, iters / 8);   // warm
    (void)hipEventRecord(e0);
    if (GAP == -2) burnR<<<cus, 256>>>(in, cyc, sink, iters); else if (GAP == -3) burn16R<<<cus, 256>>>(in, cyc, sink, iters); else if (GAP < 0) burn16<<<cus, 256>>>(in, cyc, sink, iters); else burn<(GAP < 0 ? 0 : GAP)><<<cus, 256>>>(in, cyc, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = (unsigned long long*)malloc(cus * 8);
    (void)hipMemcpy(h, cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    const double n_mfma = (double)iters * ((GAP == -1 || GAP == -3) ? 64 : 32);
    const double flops = n_mfma * 2.0 * ((GAP == -1 || GAP == -3) ? 16 * 16 * 32 : 32 * 32 * 16) * 4 * cus;       // per wave x 4 waves x CUs
    printf("%-34s %6.1f ms  %5.1f cycles/MFMA  clock %.2f GHz  %7.1f TFLOP/s fp16 dense\n", what, ms, mean / n_mfma, mean / ms / 1e6,
           flops / ms / 1e9);
    free(h); (void)hipFree(cyc); (void)hipFree(sink);
}

int main() {
    _Float16 hz[4096], hr[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) { hz[i] = (_Float16)0.f; hr[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f); }
    _Float16 *dz, *dr;
    (void)hipMalloc(&dz, sizeof(hz)); (void)hipMalloc(&dr, sizeof(hr));
    (void)hipMemcpy(dz, hz, sizeof(hz), hipMemcpyHostToDevice); (void)hipMemcpy(dr, hr, sizeof(hr), hipMemcpyHostToDevice);
    const int iters = 60000;      // x 32 MFMAs x 32 cycles ~ 61 M cycles ~ 30 ms
    run<0>("MFMA only, zero operands", dz, iters);
    run<0>("MFMA only, random operands", dr, iters);
    run<0>("MFMA only, random operands (again)", dr, iters);
    run<2>("MFMA + 2 v_fma_f32, random", dr, iters);
    run<4>("MFMA + 4 v_fma_f32, random", dr, iters);
    run<-1>("16x16x32 MFMA only, zero operands", dz, iters);
    run<-1>("16x16x32 MFMA only, random operands", dr, iters);
    run<-2>("32x32x16, random ROTATING operands", dr, iters);
    run<-3>("16x16x32, random ROTATING operands", dr, iters);
    run<0>("MFMA only, random operands, 4x longer", dr, iters * 4);
    for (int rep = 0; rep < 2; ++rep) {
        runU<0>("unit hl: Al.Bh Ah.Bl Ah.Bh (prod.)", dr, iters * 4 / 3);
        runU<1>("unit hl: Al.Bh Ah.Bh Ah.Bl", dr, iters * 4 / 3);
        runU<2>("unit hl: mirrored odd tiles", dr, iters * 4 / 3);
    }
    return 0;
}
