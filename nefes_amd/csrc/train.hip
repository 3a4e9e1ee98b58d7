// Weight-gradient (training) pass of the field MLP -- SURVEY.md §8f row 3.
// What autograd does in the reference for `loss.backward()` through NeRFH_NFF.forward when the NeRF weights
// require grad (script/run_nefes.py:42-108 -> models/nerfh_nff.py:525-576), restated layer by layer on
// v_mfma_f32_32x32x2_f32 over the tile-major buffers the TRAIN instances of field_fwd_kernel write
// (layout.h: buf[tile128][row][128 samples]):
//   train_head_grad_kernel : d raw -> head pre-activation gradients (softplus' = 1-exp(-y), sigmoid' = y(1-y))
//   train_dx_kernel<NT>    : G_in[i][s] (+)= sum_o W[o][i] G_out[o][s], then ReLU' from the saved pre-activation
//   train_dw_kernel<..>    : dW[o][i] = sum_s G_out[o][s] * act(X[i][s])   (split over sample tiles, partials summed
//                            by the caller in a fixed order: deterministic)
// Operands are read straight from global memory (L2-resident weights, streaming activations), 16 bytes per lane =
// four k-steps per load; no LDS.  Roofline: MFMA-bound like the forward (same MACs per sample for dX and for dW);
// this first version is bounded by L2->register operand traffic instead (see DESIGN.md §4.5 for measured rates).
#include "../../include/nefes_hip.h"
#include "layout.h"
#include <hip/hip_runtime.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void train_head_grad_kernel(int W, int C, int full, long long M, int S, int R, int rows,
                                                              const float* __restrict__ raw_t, const float* __restrict__ g_raw_t,
                                                              float* __restrict__ dacts) {
    const int tile = blockIdx.x, s = threadIdx.x;
    const long long m = (long long)tile * 128 + s;
    const bool valid = m < M;
    const int ray = valid ? (int)(m / S) : 0, smp = valid ? (int)(m - (long long)ray * S) : 0;
    const size_t col = (size_t)ray * R * S + smp;
    float* base = dacts + (size_t)tile * rows * 128 + s;
    const int C3 = 3 + C, ntr = (C3 + 31) / 32;
    float* rgb = base + (size_t)nefes_train_row(W, C, NEFES_TB_RGB) * 128;
    for (int c = 0; c < 32 * ntr; ++c) rgb[(size_t)c * 128] = (valid && c < C3) ? g_raw_t[col + (size_t)c * S] : 0.f;
    float* sig = base + (size_t)nefes_train_row(W, C, NEFES_TB_SIG) * 128;
    float ds = 0.f;
    if (valid) ds = g_raw_t[col + (size_t)C3 * S] * (1.f - expf(-raw_t[col + (size_t)C3 * S]));
    for (int c = 0; c < 32; ++c) sig[(size_t)c * 128] = c == 0 ? ds : 0.f;
    if (full) {
        float* th = base + (size_t)nefes_train_row(W, C, NEFES_TB_TH) * 128;
        for (int c = 0; c < 32; ++c) {
            float v = 0.f;
            if (valid && c < 5) {
                const float y = raw_t[col + (size_t)(C3 + 1 + c) * S], g = g_raw_t[col + (size_t)(C3 + 1 + c) * S];
                v = c < 3 ? g * (y * (1.f - y)) : g * (1.f - expf(-y));
            }
            th[(size_t)c * 128] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One wave = 32 samples x NT*32 input rows.  A operand: Wt[i][o] (transposed weights, row-major, ld = ldw), four o's per
// 16-byte load; B operand: G[o][sample], one dword per o (two 128-byte row segments per instruction).
template <int NT>
__global__ __launch_bounds__(256, 2) void train_dx_kernel(int rows, const float* __restrict__ gbuf, int g_row0, int n_out,
                                                          const float* __restrict__ wt, int ldw, const float* __restrict__ acts,
                                                          int dst_row0, int accumulate, int mask, float* dbuf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const size_t tile_off = (size_t)blockIdx.x * rows * 128;
    const float* g = gbuf + tile_off + (size_t)g_row0 * 128 + wave * 32 + j;
    const float* wrow = wt + (size_t)j * ldw + 4 * kh;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // operands of step q+1 are requested before the MFMAs of step q issue (the G rows come from HBM)
    const int nq = n_out / 8;
    float4 a[NT];
    float b[4];
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = *(const float4*)(wrow + (size_t)32 * t * ldw);
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = g[(size_t)(4 * kh + c) * 128];
    for (int q = 0; q < nq; ++q) {
        float4 an[NT];
        float bn[4];
        const int qn = q + 1 < nq ? q + 1 : q;
#pragma unroll
        for (int t = 0; t < NT; ++t) an[t] = *(const float4*)(wrow + (size_t)32 * t * ldw + 8 * qn);
#pragma unroll
        for (int c = 0; c < 4; ++c) bn[c] = g[(size_t)(8 * qn + 4 * kh + c) * 128];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = mfma2(a[t].x, b[0], acc[t]);
            acc[t] = mfma2(a[t].y, b[1], acc[t]);
            acc[t] = mfma2(a[t].z, b[2], acc[t]);
            acc[t] = mfma2(a[t].w, b[3], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = an[t];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = bn[c];
    }
    const size_t o0 = tile_off + (size_t)(dst_row0 + 4 * kh) * 128 + wave * 32 + j;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t o = o0 + (size_t)(32 * t + nefes_rho(0, r)) * 128;
            float v = acc[t][r];
            if (accumulate) v += dbuf[o];
            if (mask) v = acts[o] > 0.f ? v : 0.f;
            dbuf[o] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------------
// One wave = a (32 NTO) x (32 NTI) block of dW over its range of sample tiles.  A operand: G[o][sample], B operand:
// X[i][sample]; both 16 bytes per lane = four k-steps (k = sample) per load.
template <int NTO, int NTI>
__global__ __launch_bounds__(64) void train_dw_kernel(int n_tiles, int rows, const float* __restrict__ gbuf, int g_row0,
                                                      const float* __restrict__ xbuf, int x_row0, int x_relu, int in_blocks,
                                                      int splits, int n_in_pad, int n_out_pad, float* __restrict__ partial) {
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    const int ib = blockIdx.x % in_blocks, ob = blockIdx.x / in_blocks, sp = blockIdx.y;
    const int t_lo = (int)((long long)n_tiles * sp / splits), t_hi = (int)((long long)n_tiles * (sp + 1) / splits);
    f32x16 acc[NTO][NTI];
#pragma unroll
    for (int to = 0; to < NTO; ++to)
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[to][ti][r] = 0.f;
    const size_t go = (size_t)(g_row0 + 32 * NTO * ob + m) * 128 + 4 * kh;
    const size_t xo = (size_t)(x_row0 + 32 * NTI * ib + m) * 128 + 4 * kh;
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const float* g = gbuf + (size_t)tile * rows * 128 + go;
        const float* x = xbuf + (size_t)tile * rows * 128 + xo;
        // one 128-byte line of every row per step (4 x 16 bytes per lane issued together, so each line is fetched once)
#pragma unroll 1
        for (int line = 0; line < 4; ++line) {
            float4 a[NTO][4], b[NTI][4];
#pragma unroll
            for (int to = 0; to < NTO; ++to)
#pragma unroll
                for (int u = 0; u < 4; ++u) a[to][u] = *(const float4*)(g + (size_t)32 * to * 128 + 32 * line + 8 * u);
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = *(const float4*)(x + (size_t)32 * ti * 128 + 32 * line + 8 * u);
                    if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    b[ti][u] = v;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int to = 0; to < NTO; ++to)
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti) {
                        acc[to][ti] = mfma2(a[to][u].x, b[ti][u].x, acc[to][ti]);
                        acc[to][ti] = mfma2(a[to][u].y, b[ti][u].y, acc[to][ti]);
                        acc[to][ti] = mfma2(a[to][u].z, b[ti][u].z, acc[to][ti]);
                        acc[to][ti] = mfma2(a[to][u].w, b[ti][u].w, acc[to][ti]);
                    }
        }
    }
    float* out = partial + (size_t)sp * n_out_pad * n_in_pad + (size_t)(32 * NTO * ob + 4 * kh) * n_in_pad + 32 * NTI * ib + m;
#pragma unroll
    for (int to = 0; to < NTO; ++to)
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(32 * to + nefes_rho(0, r)) * n_in_pad + 32 * ti] = acc[to][ti][r];
}

template <int NTO, int NTI>
int launch_dw(int n_tiles, int rows, const float* g, int g_row0, int out_tiles, const float* x, int x_row0, int in_tiles,
              int x_relu, int splits, float* partial, hipStream_t st) {
    const int ob = out_tiles / NTO, ib = in_tiles / NTI;
    train_dw_kernel<NTO, NTI><<<dim3((unsigned)(ob * ib), (unsigned)splits), dim3(64), 0, st>>>(
        n_tiles, rows, g, g_row0, x, x_row0, x_relu, ib, splits, 32 * in_tiles, 32 * out_tiles, partial);
    return (int)hipGetLastError();
}

}   // namespace

extern "C" int nefes_train_head_grad(const NefesNetDesc* desc, int mode, int N, int S, const float* raw_t,
                                     const float* g_raw_t, float* dacts, void* stream) {
    if (!desc || !raw_t || !g_raw_t || !dacts || N <= 0 || S <= 0) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_STATIC && mode != NEFES_FIELD_FULL) return NEFES_E_UNSUPPORTED;
    if (mode == NEFES_FIELD_FULL && !desc->has_transient) return NEFES_E_BADARG;
    const long long M = (long long)N * S;
    const int n_tiles = (int)((M + 127) / 128), C = desc->feat_dim;
    const int R = mode == NEFES_FIELD_STATIC ? 3 + C + 1 : 3 + C + 6;
    train_head_grad_kernel<<<dim3((unsigned)n_tiles), dim3(128), 0, (hipStream_t)stream>>>(
        desc->width, C, mode == NEFES_FIELD_FULL, M, S, R, nefes_train_row(desc->width, C, NEFES_TB_END), raw_t, g_raw_t, dacts);
    return (int)hipGetLastError();
}

extern "C" int nefes_train_dx(int64_t n_tiles, int rows, const float* dacts_in, int g_row0, int n_out, const float* wt,
                              int ldw, int n_in, const float* acts, int dst_row0, int accumulate, int mask, float* dacts_out,
                              void* stream) {
    if (n_tiles <= 0 || rows <= 0 || !dacts_in || !wt || !dacts_out || (mask && !acts)) return NEFES_E_BADARG;
    if (n_out <= 0 || n_out % 8 || ldw < n_out || ldw % 4 || g_row0 < 0 || dst_row0 < 0) return NEFES_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_in == 256)
        train_dx_kernel<8><<<dim3((unsigned)n_tiles), dim3(256), 0, st>>>(rows, dacts_in, g_row0, n_out, wt, ldw, acts, dst_row0, accumulate, mask, dacts_out);
    else if (n_in == 128)
        train_dx_kernel<4><<<dim3((unsigned)n_tiles), dim3(256), 0, st>>>(rows, dacts_in, g_row0, n_out, wt, ldw, acts, dst_row0, accumulate, mask, dacts_out);
    else if (n_in == 64)
        train_dx_kernel<2><<<dim3((unsigned)n_tiles), dim3(256), 0, st>>>(rows, dacts_in, g_row0, n_out, wt, ldw, acts, dst_row0, accumulate, mask, dacts_out);
    else
        return NEFES_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

extern "C" int nefes_train_dw(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts,
                              int x_row0, int n_in, int x_relu, int splits, float* partial, void* stream) {
    if (n_tiles <= 0 || rows <= 0 || !dacts || !acts || !partial || splits <= 0 || splits > n_tiles) return NEFES_E_BADARG;
    if (n_out <= 0 || n_out % 32 || n_in <= 0 || n_in % 32 || g_row0 < 0 || x_row0 < 0) return NEFES_E_BADARG;
    const int ot = n_out / 32, it = n_in / 32;
    hipStream_t st = (hipStream_t)stream;
    const int nto = ot % 2 == 0 ? 2 : 1, nti = it % 4 == 0 ? 4 : (it % 2 == 0 ? 2 : 1);
#define NEFES_DW(O, I) \
    if (nto == O && nti == I) return launch_dw<O, I>((int)n_tiles, rows, dacts, g_row0, ot, acts, x_row0, it, x_relu, splits, partial, st);
    NEFES_DW(2, 4) NEFES_DW(2, 2) NEFES_DW(2, 1) NEFES_DW(1, 4) NEFES_DW(1, 2) NEFES_DW(1, 1)
#undef NEFES_DW
    return NEFES_E_UNSUPPORTED;
}
