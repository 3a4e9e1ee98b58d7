// Hardware semantics the fp16 two-part kernels rely on (csrc/field_h3.h), checked in isolation:
//   1. v_fma_mixlo/mixhi_f16 split: hi = RNE_f16(x r), lo = RNE_f16(x r - hi); hi + lo == x r to 2^-22 |x r|, subnormal lo kept;
//   2. v_mfma_f32_32x32x16_f16 consumes subnormal fp16 operands (no flush);
//   3. v_permlane32_swap_b32 of a register with itself: lower-half values on both halves / upper-half values on both halves.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe/h3_probe.hip -o /tmp/h3_probe && /tmp/h3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void split_k(const float* x, float r, float* hi, float* lo, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    uint32_t h, l;
    asm volatile(
        "v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1), "v"(r));
    _Float16 hh[2], ll[2];
    __builtin_memcpy(hh, &h, 4); __builtin_memcpy(ll, &l, 4);
    hi[2 * i] = (float)hh[0]; hi[2 * i + 1] = (float)hh[1];
    lo[2 * i] = (float)ll[0]; lo[2 * i + 1] = (float)ll[1];
}

__global__ void mfma_subnormal_k(float* out) {
    // A = all 2^-20 (fp16 subnormal), B = all 2^10: D[i][j] = 16 * 2^-10 = 2^-6 if subnormals are consumed, 0 if flushed
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1024.f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

__global__ void swap_k(uint32_t* out) {
    uint32_t u = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}

int main() {
    const int n = 1 << 16;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float m = (float)rand() / RAND_MAX * 2.f - 1.f;
        x[i] = ldexpf(m, (rand() % 40) - 30);      // magnitudes 2^-30 .. 2^10
    }
    float *dx, *dh, *dl;
    hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    const float r = 16.f;                           // scaled values up to 2^14
    split_k<<<n / 2 / 256, 256>>>(dx, r, dh, dl, n);
    std::vector<float> hi(n), lo(n);
    hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_small = 0;
    int bad_rne = 0;
    for (int i = 0; i < n; ++i) {
        const double y = (double)x[i] * r, rec = (double)hi[i] + (double)lo[i];
        const double err = fabs(rec - y);
        if (fabs(y) >= ldexp(1.0, -3)) worst = fmax(worst, err / fabs(y));          // full two-part precision expected: <= 2^-22
        else worst_small = fmax(worst_small, err);                                   // below: absolute error <= 2^-25 (subnormal lo)
        if ((float)(_Float16)(float)y != hi[i]) ++bad_rne;                           // hi is the round-to-nearest fp16 of x r
    }
    printf("split: worst rel err (|y| >= 2^-3) %.3e (2^-22 = %.3e); worst abs err below %.3e (2^-25 = %.3e); hi != RNE: %d\n",
           worst, ldexp(1.0, -22), worst_small, ldexp(1.0, -25), bad_rne);
    float* dout; hipMalloc(&dout, 4);
    mfma_subnormal_k<<<1, 64>>>(dout);
    float o; hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost);
    printf("mfma subnormal operand: D = %.6e (consumed: %.6e, flushed: 0)\n", o, ldexp(1.0, -6));
    uint32_t* ds; hipMalloc(&ds, 128 * 4);
    swap_k<<<1, 64>>>(ds);
    uint32_t s[128]; hipMemcpy(s, ds, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; ++i) ok &= (s[i] == (uint32_t)(i & 31)) && (s[64 + i] == (uint32_t)(32 + (i & 31)));
    printf("permlane32_swap(u,u): r0 = lower half on both halves, r1 = upper half on both halves: %s (r0[5]=%u r0[37]=%u r1[5]=%u r1[37]=%u)\n",
           ok ? "yes" : "NO", s[5], s[37], s[64 + 5], s[64 + 37]);
    const int pass = worst <= ldexp(1.0, -21.9) && worst_small <= ldexp(1.0, -24.9) && bad_rne == 0 && o == ldexpf(1.f, -6) && ok;
    printf("%s\n", pass ? "H3 PROBE PASS" : "H3 PROBE FAIL");
    return pass ? 0 : 1;
}
