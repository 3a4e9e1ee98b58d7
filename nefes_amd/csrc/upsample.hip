// Bicubic up-sampling of the fused feature image and its backward (refinement loop, after the render path).
// Restates what `torch.nn.Upsample(size=(H, W), mode='bicubic')` does at script/dm/DFM_APR_refine.py:114,118
// (ATen upsample_bicubic2d, align_corners=False: src = scale*(dst+0.5)-0.5 unclamped, taps floor(src)-1..+2 clamped
// to the image, cubic-convolution weights with A = -0.75, rows first then columns).
// Why a kernel: the library backward scatters 16 atomics per output pixel and costs 4.7 ms per 128x240x320 gradient --
// 40 % of a refinement iteration at 80x60.  Here the backward is a separable GATHER (rows, then columns): every input
// pixel sums the output pixels whose clamped taps hit it, no atomics, deterministic.  HBM-bound: 39 MB read per call.
#include "../../include/nefes_hip.h"
#include <hip/hip_runtime.h>

#include "bicubic.h"

namespace {
using namespace nefes_bicubic;

// out = the window [oy0, oy0+CH) x [ox0, ox0+CW) of the up-sampled OH x OW image (the loop crops 10 pixels per side right after
// up-sampling, DFM_APR_refine.py:115,119: only the window is ever computed, stored, or differentiated)
__global__ __launch_bounds__(256) void bicubic_up_fwd_kernel(long planes, int h, int w, int CH, int CW, int oy0, int ox0, float sy, float sx,
                                                             const float* __restrict__ in, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * CH * CW) return;
    const int ox = (int)(idx % CW) + ox0, oy = (int)((idx / CW) % CH) + oy0;
    const long p = idx / ((long)CW * CH);
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
// src holds only the window [o0, o0 + n_win) of the n_out up-sampled positions along the axis.
__global__ __launch_bounds__(256) void bicubic_gather_kernel(long outer, int n_in, int n_out, int o0, int n_win, long inner, float scale,
                                                             float inv_scale, const float* __restrict__ src, float* __restrict__ dst) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= outer * n_in * inner) return;
    const long q = idx % inner;
    const int y = (int)((idx / inner) % n_in);
    const long p = idx / (inner * n_in);
    (void)n_out;
    const float acc = gather_axis(y, n_in, o0, n_win, scale, inv_scale, src + p * n_win * inner + q, inner);
    dst[idx] = acc;
}

// first / count / wt of GatherTable for one axis: one thread per source index.  Positions whose taps miss y inside [lo, hi] get
// weight 0 (the span is contiguous up to such holes at the clamped borders).
__global__ void gather_table_kernel(int n_in, int o0, int n_win, float scale, float inv_scale, int T, int* first, int* count, float* wt) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= n_in) return;
    int lo = (int)floorf(((float)y - 1.5f) * inv_scale - 0.5f) - 1;
    int hi = (int)ceilf(((float)y + 2.5f) * inv_scale - 0.5f) + 1;
    lo = lo < o0 ? o0 : lo;
    hi = hi > o0 + n_win - 1 ? o0 + n_win - 1 : hi;
    int f = -1, n = 0;
    for (int o = lo; o <= hi; ++o) {
        const Taps t = taps_of(scale, o);
        float wsum = 0.f;
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (clampi(t.base - 1 + k, n_in) == y) { wsum += t.w[k]; hit = true; }
        if (hit && f < 0) f = o;
        if (f >= 0 && o - f < T) {
            wt[(long)y * T + (o - f)] = hit ? wsum : 0.f;
            if (hit) n = o - f + 1;
        }
    }
    first[y] = f < 0 ? 0 : f - o0;
    count[y] = n;
}

}   // namespace

extern "C" int nefes_bicubic_gather_table(int n_in, int n_out, int o0, int n_win, int T, int* first, int* count, float* wt, void* stream_) {
    if (n_in <= 0 || n_out <= 0 || o0 < 0 || n_win <= 0 || o0 + n_win > n_out || T <= 0 || !first || !count || !wt) return NEFES_E_BADARG;
    if (T < (int)(4.f * n_out / n_in) + 8) return NEFES_E_BADARG;         // a source index is hit from at most ~4 scale + a few positions
    gather_table_kernel<<<dim3((n_in + 63) / 64), dim3(64), 0, (hipStream_t)stream_>>>(n_in, o0, n_win, (float)n_in / n_out, (float)n_out / n_in, T, first, count, wt);
    return (int)hipGetLastError();
}

static bool window_ok(int O, int o0, int n) { return o0 >= 0 && n > 0 && o0 + n <= O; }

extern "C" int nefes_bicubic_up_fwd(int64_t planes, int h, int w, int OH, int OW, int oy0, int ox0, int CH, int CW, const float* in,
                                    float* out, void* stream_) {
    if (planes < 0 || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || !window_ok(OH, oy0, CH) || !window_ok(OW, ox0, CW) || !in || !out)
        return NEFES_E_BADARG;
    const long n = (long)planes * CH * CW;
    if (n == 0) return 0;
    bicubic_up_fwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(planes, h, w, CH, CW, oy0, ox0, (float)h / OH, (float)w / OW, in, out);
    return (int)hipGetLastError();
}

extern "C" int nefes_bicubic_up_bwd(int64_t planes, int h, int w, int OH, int OW, int oy0, int ox0, int CH, int CW, const float* g_out,
                                    float* tmp, float* g_in, void* stream_) {
    if (planes < 0 || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || !window_ok(OH, oy0, CH) || !window_ok(OW, ox0, CW) || !g_out || !tmp ||
        !g_in)
        return NEFES_E_BADARG;
    if (planes == 0) return 0;
    // rows: tmp[p][y][cx] = sum_{oy in window} Wy(oy->y) g_out[p][oy - oy0][cx]                       (tmp: [planes, h, CW])
    long n1 = (long)planes * h * CW;
    bicubic_gather_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(planes, h, OH, oy0, CH, CW, (float)h / OH, (float)OH / h, g_out, tmp);
    // columns: g_in[p*h + y][x] = sum_{ox in window} Wx(ox->x) tmp[p*h + y][ox - ox0]
    long n2 = (long)planes * h * w;
    bicubic_gather_kernel<<<dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>((long)planes * h, w, OW, ox0, CW, 1, (float)w / OW, (float)OW / w, tmp, g_in);
    return (int)hipGetLastError();
}
