// Candidate instruction sequences for the fp16 two-part split of a pair of values (field_h3.h split_pair_h): cycles per pair with
// the vector ALU alone (one wave per SIMD), and bit-identity of the candidates' (hi, lo) words with the production sequence.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/split_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// production: four v_fma_mix
__device__ __forceinline__ void split_cur(uint32_t& h, uint32_t& l, float x0, float x1, float r) {
    asm volatile(
        "v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(x0), "v"(x1), "v"(r));
}
// candidate B: two multiplies, packed convert, two mixed fmas (y - hi) -> lo
__device__ __forceinline__ void split_b(uint32_t& h, uint32_t& l, float x0, float x1, float r) {
    float y0, y1;
    const float one = 1.f;
    asm volatile(
        "v_mul_f32 %2, %4, %6\n\t"
        "v_mul_f32 %3, %5, %6\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, %7, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l), "=&v"(y0), "=&v"(y1)
        : "v"(x0), "v"(x1), "v"(r), "v"(one));
}
// candidate C: two multiplies, packed convert, two unpacking converts, two subtracts, packed convert
__device__ __forceinline__ void split_c(uint32_t& h, uint32_t& l, float x0, float x1, float r) {
    float y0, y1, t0, t1;
    asm volatile(
        "v_mul_f32 %2, %6, %8\n\t"
        "v_mul_f32 %3, %7, %8\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_cvt_f32_f16 %4, %0\n\t"
        "v_cvt_f32_f16_sdwa %5, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
        "v_sub_f32 %4, %2, %4\n\t"
        "v_sub_f32 %5, %3, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %4, %5"
        : "=&v"(h), "=&v"(l), "=&v"(y0), "=&v"(y1), "=&v"(t0), "=&v"(t1)
        : "v"(x0), "v"(x1), "v"(r));
}
// candidate D: like C with packed multiply / subtract, left to the compiler (vector types: v_pk_mul_f32, v_cvt_pk_f16_f32,
// v_cvt_f32_f16 x 2, v_pk_add_f32, v_cvt_pk_f16_f32)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_d(uint32_t& h, uint32_t& l, float x0, float x1, float r) {
    const f32x2 x = {x0, x1};
    const f32x2 y = x * r;
    const f16x2 hh = __builtin_convertvector(y, f16x2);
    const f32x2 d = y - __builtin_convertvector(hh, f32x2);
    const f16x2 ll = __builtin_convertvector(d, f16x2);
    __builtin_memcpy(&h, &hh, 4);
    __builtin_memcpy(&l, &ll, 4);
}

template <int KIND>
__device__ __forceinline__ void split(uint32_t& h, uint32_t& l, float x0, float x1, float r) {
    if (KIND == 0) split_cur(h, l, x0, x1, r);
    if (KIND == 1) split_b(h, l, x0, x1, r);
    if (KIND == 2) split_c(h, l, x0, x1, r);
    if (KIND == 3) split_d(h, l, x0, x1, r);
}

template <int KIND>
__global__ __launch_bounds__(256) void timing(unsigned long long* out, uint32_t* sink) {
    float a = threadIdx.x * 0.37f + 1.f, b = a * 1.7f, r = 4.f;
    uint32_t acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            uint32_t h, l;
            asm volatile("" : "+v"(a), "+v"(b));             // (opaque: nothing is hoisted out of the loop)
            split<KIND>(h, l, a, b, r);
            asm volatile("" ::"v"(h), "v"(l));
            acc ^= h;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = t1 - t0;
    sink[threadIdx.x] = acc;
}

template <int KIND>
__global__ void values(const float* x, const float* r, uint32_t* h, uint32_t* l, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) split<KIND>(h[i], l[i], x[2 * i], x[2 * i + 1], r[i]);
}

int main() {
    unsigned long long* out; uint32_t* sink;
    (void)hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    timing<0><<<256, 256>>>(out, sink); timing<1><<<256, 256>>>(out, sink); timing<2><<<256, 256>>>(out, sink); timing<3><<<256, 256>>>(out, sink);
    unsigned long long t[4];
    hipMemcpy(t, out, 32, hipMemcpyDeviceToHost);
    const char* names[4] = {"production: 4 x v_fma_mix", "B: 2 mul, cvt_pk, 2 fma_mix", "C: 2 mul, cvt_pk, 2 cvt, 2 sub, cvt_pk", "D: pk_mul, cvt_pk, 2 cvt, pk_add, cvt_pk"};
    for (int k = 0; k < 4; ++k) printf("%-44s %.1f cycles per pair (vector ALU alone)\n", names[k], (double)t[k] / (64.0 * 32));
    // bit identity on random values over the exponent range the kernels see, subnormal results included
    const int n = 1 << 20;
    float *hx = (float*)malloc(8 * n), *hr = (float*)malloc(4 * n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 40 - 30;                            // x 2^e with the maximum brought to [2^14, 2^15)
        hr[i] = ldexpf(1.f, 14 - (e > -24 ? e : -24) + (rand() % 3 == 0 ? rand() % 12 : 0) * 0);
        for (int k = 0; k < 2; ++k) {
            float m = (float)rand() / RAND_MAX * 2.f - 1.f;
            if (rand() % 16 == 0) m = 0.f;
            hx[2 * i + k] = ldexpf(m, (e > -24 ? e : -24) - (rand() % 30));     // up to 30 binades below the sample's maximum
        }
    }
    float *dx, *dr; uint32_t *dh, *dl;
    hipMalloc(&dx, 8 * n); hipMalloc(&dr, 4 * n); hipMalloc(&dh, 4 * n); hipMalloc(&dl, 4 * n);
    hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice); hipMemcpy(dr, hr, 4 * n, hipMemcpyHostToDevice);
    uint32_t* ref_h = (uint32_t*)malloc(4 * n); uint32_t* ref_l = (uint32_t*)malloc(4 * n);
    uint32_t* got_h = (uint32_t*)malloc(4 * n); uint32_t* got_l = (uint32_t*)malloc(4 * n);
    values<0><<<n / 256, 256>>>(dx, dr, dh, dl, n);
    hipMemcpy(ref_h, dh, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(ref_l, dl, 4 * n, hipMemcpyDeviceToHost);
    for (int kind = 1; kind < 4; ++kind) {
        if (kind == 1) values<1><<<n / 256, 256>>>(dx, dr, dh, dl, n);
        if (kind == 2) values<2><<<n / 256, 256>>>(dx, dr, dh, dl, n);
        if (kind == 3) values<3><<<n / 256, 256>>>(dx, dr, dh, dl, n);
        hipMemcpy(got_h, dh, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(got_l, dl, 4 * n, hipMemcpyDeviceToHost);
        long bad = 0, badz = 0;
        for (int i = 0; i < n; ++i) {
            if (got_h[i] != ref_h[i] || got_l[i] != ref_l[i]) {
                // -0 vs +0 halves are the same value to the MFMA
                auto same = [](uint32_t a, uint32_t b) { return ((a ^ b) & 0x7fff7fffu) == 0 && ((((a | b) & 0x7fffu) == 0) || ((a ^ b) & 0x8000u) == 0) && ((((a | b) & 0x7fff0000u) == 0) || ((a ^ b) & 0x80000000u) == 0); };
                if (same(got_h[i], ref_h[i]) && same(got_l[i], ref_l[i])) ++badz; else { if (bad < 4) printf("  x = %a %a r = %a: ref %08x %08x got %08x %08x\n", hx[2 * i], hx[2 * i + 1], hr[i], ref_h[i], ref_l[i], got_h[i], got_l[i]); ++bad; }
            }
        }
        printf("%-44s differing pairs: %ld (+ %ld that differ in the sign of a zero only) of %d\n", names[kind], bad, badz, n);
    }
    return 0;
}
