// Bicubic up-sampling of the fused feature image and its backward (refinement loop, after the render path).
// Restates what `torch.nn.Upsample(size=(H, W), mode='bicubic')` does at script/dm/DFM_APR_refine.py:114,118
// (ATen upsample_bicubic2d, align_corners=False: src = scale*(dst+0.5)-0.5 unclamped, taps floor(src)-1..+2 clamped
// to the image, cubic-convolution weights with A = -0.75, rows first then columns).
// Why a kernel: the library backward scatters 16 atomics per output pixel and costs 4.7 ms per 128x240x320 gradient --
// 40 % of a refinement iteration at 80x60.  Here the backward is a separable GATHER (rows, then columns): every input
// pixel sums the output pixels whose clamped taps hit it, no atomics, deterministic.  HBM-bound: 39 MB read per call.
#include "../../include/nefes_hip.h"
#include <hip/hip_runtime.h>

namespace {

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

__global__ __launch_bounds__(256) void bicubic_up_fwd_kernel(long planes, int h, int w, int OH, int OW, float sy, float sx,
                                                             const float* __restrict__ in, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * OH * OW) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH);
    const long p = idx / ((long)OW * OH);
    const Taps ty = taps_of(sy, oy), tx = taps_of(sx, ox);
    const float* src = in + p * h * w;
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = clampi(tx.base - 1 + j, w);
    float rows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* r = src + (long)clampi(ty.base - 1 + i, h) * w;
        rows[i] = r[xs[0]] * tx.w[0] + r[xs[1]] * tx.w[1] + r[xs[2]] * tx.w[2] + r[xs[3]] * tx.w[3];
    }
    out[idx] = rows[0] * ty.w[0] + rows[1] * ty.w[1] + rows[2] * ty.w[2] + rows[3] * ty.w[3];
}

// One gather pass along one axis.  src is [planes, n_out_axis, inner] (AXIS_ROWS) or [planes*rows, n_out_axis] (columns):
// generalised as dst[p][y][q] = sum_o W(o -> y) src[p][o][q] with strides given.
__global__ __launch_bounds__(256) void bicubic_gather_kernel(long outer, int n_in, int n_out, long inner, float scale,
                                                             float inv_scale, const float* __restrict__ src, float* __restrict__ dst) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= outer * n_in * inner) return;
    const long q = idx % inner;
    const int y = (int)((idx / inner) % n_in);
    const long p = idx / (inner * n_in);
    int lo = (int)floorf(((float)y - 1.5f) * inv_scale - 0.5f) - 1;
    int hi = (int)ceilf(((float)y + 2.5f) * inv_scale - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
    const float* s = src + p * n_out * inner + q;
    float acc = 0.f;
    for (int o = lo; o <= hi; ++o) {
        const Taps t = taps_of(scale, o);
        float wsum = 0.f;
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (clampi(t.base - 1 + k, n_in) == y) { wsum += t.w[k]; hit = true; }
        if (hit) acc += wsum * s[(long)o * inner];
    }
    dst[idx] = acc;
}

}   // namespace

extern "C" int nefes_bicubic_up_fwd(int64_t planes, int h, int w, int OH, int OW, const float* in, float* out, void* stream_) {
    if (planes < 0 || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || !in || !out) return NEFES_E_BADARG;
    const long n = (long)planes * OH * OW;
    if (n == 0) return 0;
    bicubic_up_fwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(planes, h, w, OH, OW, (float)h / OH, (float)w / OW, in, out);
    return (int)hipGetLastError();
}

extern "C" int nefes_bicubic_up_bwd(int64_t planes, int h, int w, int OH, int OW, const float* g_out, float* tmp, float* g_in,
                                    void* stream_) {
    if (planes < 0 || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || !g_out || !tmp || !g_in) return NEFES_E_BADARG;
    if (planes == 0) return 0;
    // rows: tmp[p][y][ox] = sum_oy Wy(oy->y) g_out[p][oy][ox]
    long n1 = (long)planes * h * OW;
    bicubic_gather_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(planes, h, OH, OW, (float)h / OH, (float)OH / h, g_out, tmp);
    // columns: g_in[p*h + y][x] = sum_ox Wx(ox->x) tmp[p*h + y][ox]
    long n2 = (long)planes * h * w;
    bicubic_gather_kernel<<<dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>((long)planes * h, w, OW, 1, (float)w / OW, (float)OW / w, tmp, g_in);
    return (int)hipGetLastError();
}
