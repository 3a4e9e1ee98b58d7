// Bicubic taps of ATen's upsample_bicubic2d (align_corners=False, A = -0.75): shared by upsample.hip and refine.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nefes_bicubic {

__device__ __forceinline__ float cc1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }

struct Taps {
    int base;        // floor(src): taps are base-1 .. base+2 (to be clamped)
    float w[4];
};

__device__ __forceinline__ Taps taps_of(float scale, int dst) {
    const float src = scale * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const float t = src - fl;
    Taps r;
    r.base = (int)fl;
    r.w[0] = cc2(t + 1.f);
    r.w[1] = cc1(t);
    const float u = 1.f - t;
    r.w[2] = cc1(u);
    r.w[3] = cc2(u + 1.f);
    return r;
}

__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

// One gather pass of the backward along one axis: dst[p][y][q] = sum over the up-sampled positions o of the window [o0, o0 + n_win)
// whose clamped taps hit y of  W(o -> y) src[p][o - o0][q].
__device__ __forceinline__ float gather_axis(int y, int n_in, int o0, int n_win, float scale, float inv_scale, const float* s, long stride) {
    int lo = (int)floorf(((float)y - 1.5f) * inv_scale - 0.5f) - 1;
    int hi = (int)ceilf(((float)y + 2.5f) * inv_scale - 0.5f) + 1;
    lo = lo < o0 ? o0 : lo;
    hi = hi > o0 + n_win - 1 ? o0 + n_win - 1 : hi;
    float acc = 0.f;
    for (int o = lo; o <= hi; ++o) {
        const Taps t = taps_of(scale, o);
        float wsum = 0.f;
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (clampi(t.base - 1 + k, n_in) == y) { wsum += t.w[k]; hit = true; }
        if (hit) acc += wsum * s[(long)(o - o0) * stride];
    }
    return acc;
}

// The same sum from a table built once per geometry (nefes_bicubic_gather_table): for source index y the window positions
// first[y] .. first[y] + count[y] - 1 (relative to o0) with weights wt[y * T + i] -- the taps depend on the sizes only, and every
// row, column and channel of every iteration recomputed them (floorf, four cubic polynomials and four clamps per candidate).
struct GatherTable {
    const int* first;
    const int* count;
    const float* wt;
    int T;
};
__device__ __forceinline__ float gather_axis_table(const GatherTable& g, int y, const float* s, long stride) {
    const int f = g.first[y], n = g.count[y];
    const float* w = g.wt + (long)y * g.T;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += w[i] * s[(long)(f + i) * stride];
    return acc;
}

}   // namespace nefes_bicubic
