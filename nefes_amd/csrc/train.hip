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
#include <stdlib.h>

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
    float* base = dacts + (size_t)tile * rows * 128;             // element (row, s) at base[nefes_train_off(row, s)] (layout.h)
    const int C3 = 3 + C, ntr = (C3 + 31) / 32;
    const int r_rgb = nefes_train_row(W, C, NEFES_TB_RGB), r_sig = nefes_train_row(W, C, NEFES_TB_SIG), r_th = nefes_train_row(W, C, NEFES_TB_TH);
    for (int c = 0; c < 32 * ntr; ++c) base[nefes_train_off(r_rgb + c, s)] = (valid && c < C3) ? g_raw_t[col + (size_t)c * S] : 0.f;
    float ds = 0.f;
    if (valid) ds = g_raw_t[col + (size_t)C3 * S] * (1.f - expf(-raw_t[col + (size_t)C3 * S]));
    for (int c = 0; c < 32; ++c) base[nefes_train_off(r_sig + c, s)] = c == 0 ? ds : 0.f;
    if (full) {
        for (int c = 0; c < 32; ++c) {
            float v = 0.f;
            if (valid && c < 5) {
                const float y = raw_t[col + (size_t)(C3 + 1 + c) * S], g = g_raw_t[col + (size_t)(C3 + 1 + c) * S];
                v = c < 3 ? g * (y * (1.f - y)) : g * (1.f - expf(-y));
            }
            base[nefes_train_off(r_th + c, s)] = v;
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
    const float* g = gbuf + tile_off;                            // element (row, sample) at nefes_train_off (layout.h)
    const int smp = wave * 32 + j;
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
    for (int c = 0; c < 4; ++c) b[c] = g[nefes_train_off(g_row0 + 4 * kh + c, smp)];
    for (int q = 0; q < nq; ++q) {
        float4 an[NT];
        float bn[4];
        const int qn = q + 1 < nq ? q + 1 : q;
#pragma unroll
        for (int t = 0; t < NT; ++t) an[t] = *(const float4*)(wrow + (size_t)32 * t * ldw + 8 * qn);
#pragma unroll
        for (int c = 0; c < 4; ++c) bn[c] = g[nefes_train_off(g_row0 + 8 * qn + 4 * kh + c, smp)];
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
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t o = tile_off + nefes_train_off(dst_row0 + 4 * kh + 32 * t + nefes_rho(0, r), smp);
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
                                                      int splits, int ld, int with_bias, long long split_stride, float* __restrict__ partial) {
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    const int ib = blockIdx.x % in_blocks, ob = blockIdx.x / in_blocks, sp = blockIdx.y;
    const int t_lo = (int)((long long)n_tiles * sp / splits), t_hi = (int)((long long)n_tiles * (sp + 1) / splits);
    const bool bias = with_bias && ib == 0;              // the first input block of every output block also sums G's rows
    float bsum[NTO];
    f32x16 acc[NTO][NTI];
#pragma unroll
    for (int to = 0; to < NTO; ++to) {
        bsum[to] = 0.f;
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[to][ti][r] = 0.f;
    }
    // element (row, sample) at nefes_train_off (layout.h): row block rb, 16-sample group sg -> (rb * 8 + sg) * 512 + (row % 32) * 16 + sample % 16
    const size_t go = (size_t)((g_row0 >> 5) + NTO * ob) * 4096 + m * 16;
    const size_t xo = (size_t)((x_row0 >> 5) + NTI * ib) * 4096 + m * 16;
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
                for (int u = 0; u < 4; ++u) a[to][u] = *(const float4*)(g + (size_t)to * 4096 + (2 * line + (u >> 1)) * 512 + 8 * (u & 1) + 4 * kh);
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = *(const float4*)(x + (size_t)ti * 4096 + (2 * line + (u >> 1)) * 512 + 8 * (u & 1) + 4 * kh);
                    if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    b[ti][u] = v;
                }
            if (bias) {
#pragma unroll
                for (int to = 0; to < NTO; ++to)
#pragma unroll
                    for (int u = 0; u < 4; ++u) bsum[to] += (a[to][u].x + a[to][u].y) + (a[to][u].z + a[to][u].w);
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
    float* out = partial + (size_t)sp * split_stride + (size_t)(32 * NTO * ob + 4 * kh) * ld + 32 * NTI * ib + m;
#pragma unroll
    for (int to = 0; to < NTO; ++to)
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(32 * to + nefes_rho(0, r)) * ld + 32 * ti] = acc[to][ti][r];
    if (bias) {                                          // column ld - 1 of the block's rows: sum over this split's samples
#pragma unroll
        for (int to = 0; to < NTO; ++to) {
            const float t = bsum[to] + __shfl_xor(bsum[to], 32);
            if (kh == 0) partial[(size_t)sp * split_stride + (size_t)(32 * NTO * ob + 32 * to + m) * ld + ld - 1] = t;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The same product on the bf16 pipe: G and act(X) are split exactly into three bf16 each as they are loaded (x = h + m + l:
// 24 = 3 x 8 mantissa bits, no scaling needed -- bf16 has fp32's exponent range) and a 16-sample step of a 32x32 block is
// six v_mfma_f32_32x32x16_bf16 (l.h, h.l, m.m, m.h, h.m, h.h: dropped terms 2^-24 relative) instead of eight
// v_mfma_f32_32x32x2_f32: 192 matrix-core cycles instead of 512.  Operand lane (row m, kh) holds samples 8 kh .. 8 kh + 7 of
// the step: two 16-byte loads; a step's two loads per row cover 64 bytes of its 128-byte line, the next step the other 64.
// Same C layout as the fp32 kernel, same partial-sum scheme, same epilogue.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct Tri { u32x4_t h, m, l; };
__device__ __forceinline__ bf16x8_t as_bf(u32x4_t v) { bf16x8_t r; __builtin_memcpy(&r, &v, 16); return r; }
__device__ __forceinline__ void tri_pair(Tri& o, int p, float x0, float x1) {      // truncation split, exact (field_x6.h split_pair)
    const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float e0 = x0 - __uint_as_float(b0 & 0xffff0000u), e1 = x1 - __uint_as_float(b1 & 0xffff0000u);
    const uint32_t c0 = __float_as_uint(e0), c1 = __float_as_uint(e1);
    const float f0 = e0 - __uint_as_float(c0 & 0xffff0000u), f1 = e1 - __uint_as_float(c1 & 0xffff0000u);
    o.h[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    o.m[p] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    o.l[p] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
}
__device__ __forceinline__ void tri_of(Tri& o, float4 a, float4 b, bool relu) {
    if (relu) {
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
    }
    tri_pair(o, 0, a.x, a.y); tri_pair(o, 1, a.z, a.w); tri_pair(o, 2, b.x, b.y); tri_pair(o, 3, b.z, b.w);
}
template <int NTO, int NTI>
__global__ __launch_bounds__(64) void train_dw_x6_kernel(int n_tiles, int rows, const float* __restrict__ gbuf, int g_row0,
                                                         const float* __restrict__ xbuf, int x_row0, int x_relu, int in_blocks,
                                                         int splits, int ld, int with_bias, long long split_stride, float* __restrict__ partial) {
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    const int ib = blockIdx.x % in_blocks, ob = blockIdx.x / in_blocks, sp = blockIdx.y;
    const int t_lo = (int)((long long)n_tiles * sp / splits), t_hi = (int)((long long)n_tiles * (sp + 1) / splits);
    const bool bias = with_bias && ib == 0;              // bias gradient = row sums of G: the operand is in registers anyway
    float bsum[NTO];
    f32x16 acc[NTO][NTI];
#pragma unroll
    for (int to = 0; to < NTO; ++to) {
        bsum[to] = 0.f;
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[to][ti][r] = 0.f;
    }
    // element (row, sample) at nefes_train_off (layout.h): a step's operands are whole contiguous 2 KiB blocks (32 rows x 16 samples)
    const size_t go = (size_t)((g_row0 >> 5) + NTO * ob) * 4096 + m * 16 + 8 * kh;
    const size_t xo = (size_t)((x_row0 >> 5) + NTI * ib) * 4096 + m * 16 + 8 * kh;
    // one wave per SIMD at most (256 accumulator registers for the 128 x 128 block), so the memory latency is covered by requests
    // in flight, not by other waves: two register sets, the loads of steps s + 1 and s + 2 are outstanding while the MFMAs of step s
    // issue.  A set is free as soon as its raw values are split (tri_of), i.e. at the top of its step, and is re-requested there.
    float4 ra[2][NTO][2], rb[2][NTI][2];
    auto request = [&](int tile, int grp, int set) {
        const float* g = gbuf + (size_t)tile * rows * 128 + go + 512 * grp;
        const float* x = xbuf + (size_t)tile * rows * 128 + xo + 512 * grp;
#pragma unroll
        for (int to = 0; to < NTO; ++to) { ra[set][to][0] = *(const float4*)(g + (size_t)4096 * to); ra[set][to][1] = *(const float4*)(g + (size_t)4096 * to + 4); }
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti) { rb[set][ti][0] = *(const float4*)(x + (size_t)4096 * ti); rb[set][ti][1] = *(const float4*)(x + (size_t)4096 * ti + 4); }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 bsum2[NTO];
#pragma unroll
    for (int to = 0; to < NTO; ++to) bsum2[to] = f32x2{0.f, 0.f};
    if (t_lo < t_hi) { request(t_lo, 0, 0); request(t_lo, 1, 1); }
    for (int tile = t_lo; tile < t_hi; ++tile) {
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {                    // 16 samples per step
            constexpr int kSets = 2;
            const int set = grp % kSets;
            Tri A[NTO], B[NTI];
            if (bias) {                                        // packed adds: two row-sum lanes per instruction
#pragma unroll
                for (int to = 0; to < NTO; ++to) {
                    const float4 u = ra[set][to][0], w = ra[set][to][1];
                    bsum2[to] += (f32x2{u.x, u.y} + f32x2{u.z, u.w}) + (f32x2{w.x, w.y} + f32x2{w.z, w.w});
                }
            }
#pragma unroll
            for (int to = 0; to < NTO; ++to) tri_of(A[to], ra[set][to][0], ra[set][to][1], false);
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) tri_of(B[ti], rb[set][ti][0], rb[set][ti][1], x_relu != 0);
            {
                const int ng = (grp + kSets) % 8, nt = tile + (grp + kSets) / 8;
                if (nt < t_hi) request(nt, ng, set);
            }
#pragma unroll
            for (int to = 0; to < NTO; ++to)
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti) {
                    f32x16 c = acc[to][ti];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].l), as_bf(B[ti].h), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].h), as_bf(B[ti].l), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].m), as_bf(B[ti].m), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].m), as_bf(B[ti].h), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].h), as_bf(B[ti].m), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(A[to].h), as_bf(B[ti].h), c, 0, 0, 0);
                    acc[to][ti] = c;
                }
        }
    }
#pragma unroll
    for (int to = 0; to < NTO; ++to) bsum[to] = bsum2[to].x + bsum2[to].y;
    float* out = partial + (size_t)sp * split_stride + (size_t)(32 * NTO * ob + 4 * kh) * ld + 32 * NTI * ib + m;
#pragma unroll
    for (int to = 0; to < NTO; ++to)
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(32 * to + nefes_rho(0, r)) * ld + 32 * ti] = acc[to][ti][r];
    if (bias) {                                          // column ld - 1 of the block's rows: sum over this split's samples
#pragma unroll
        for (int to = 0; to < NTO; ++to) {
            const float t = bsum[to] + __shfl_xor(bsum[to], 32);
            if (kh == 0) partial[(size_t)sp * split_stride + (size_t)(32 * NTO * ob + 32 * to + m) * ld + ld - 1] = t;
        }
    }
}

template <int NTO, int NTI>
int launch_dw(int n_tiles, int rows, const float* g, int g_row0, int out_tiles, const float* x, int x_row0, int in_tiles,
              int x_relu, int splits, int with_bias, long long split_stride, float* partial, hipStream_t st) {
    const int ld = 32 * in_tiles + (with_bias ? 1 : 0);
    if (split_stride <= 0) split_stride = (long long)32 * out_tiles * ld;      // partials of one launch back to back
    const int ob = out_tiles / NTO, ib = in_tiles / NTI;
    static const bool f32_path = [] { const char* e = getenv("NEFES_TRAIN_DW"); return e && e[0] == 'f'; }();   // "f32": the fp32-MFMA kernel
    if (f32_path)
        train_dw_kernel<NTO, NTI><<<dim3((unsigned)(ob * ib), (unsigned)splits), dim3(64), 0, st>>>(
            n_tiles, rows, g, g_row0, x, x_row0, x_relu, ib, splits, ld, with_bias, split_stride, partial);
    else
        train_dw_x6_kernel<NTO, NTI><<<dim3((unsigned)(ob * ib), (unsigned)splits), dim3(64), 0, st>>>(
            n_tiles, rows, g, g_row0, x, x_row0, x_relu, ib, splits, ld, with_bias, split_stride, partial);
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

static int train_dw_impl(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts,
                         int x_row0, int n_in, int x_relu, int splits, int with_bias, long long split_stride, float* partial, void* stream) {
    if (n_tiles <= 0 || rows <= 0 || !dacts || !acts || !partial || splits <= 0 || splits > n_tiles) return NEFES_E_BADARG;
    if (n_out <= 0 || n_out % 32 || n_in <= 0 || n_in % 32 || g_row0 < 0 || x_row0 < 0) return NEFES_E_BADARG;
    const int ot = n_out / 32, it = n_in / 32;
    hipStream_t st = (hipStream_t)stream;
    static const bool f32_dw = [] { const char* e = getenv("NEFES_TRAIN_DW"); return e && e[0] == 'f'; }();
    // bf16x6 kernel: a 128 x 128 block per wave where the shapes allow (every row of G and X is then read by ONE workgroup);
    // likewise all five tiles of the rgb+feature head (3 + 128 channels) and all four of a Wd = 128 layer over a 64-wide input
    if (!f32_dw && ot % 4 == 0 && it % 4 == 0) return launch_dw<4, 4>((int)n_tiles, rows, dacts, g_row0, ot, acts, x_row0, it, x_relu, splits, with_bias, split_stride, partial, st);
    if (!f32_dw && ot == 5 && it == 2) return launch_dw<5, 2>((int)n_tiles, rows, dacts, g_row0, ot, acts, x_row0, it, x_relu, splits, with_bias, split_stride, partial, st);
    if (!f32_dw && ot % 4 == 0 && it == 2) return launch_dw<4, 2>((int)n_tiles, rows, dacts, g_row0, ot, acts, x_row0, it, x_relu, splits, with_bias, split_stride, partial, st);
    const int nto = ot % 2 == 0 ? 2 : 1, nti = it % 4 == 0 ? 4 : (it % 2 == 0 ? 2 : 1);
#define NEFES_DW(O, I) \
    if (nto == O && nti == I) return launch_dw<O, I>((int)n_tiles, rows, dacts, g_row0, ot, acts, x_row0, it, x_relu, splits, with_bias, split_stride, partial, st);
    NEFES_DW(2, 4) NEFES_DW(2, 2) NEFES_DW(2, 1) NEFES_DW(1, 4) NEFES_DW(1, 2) NEFES_DW(1, 1)
#undef NEFES_DW
    return NEFES_E_UNSUPPORTED;
}

extern "C" int nefes_train_dw(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts,
                              int x_row0, int n_in, int x_relu, int splits, float* partial, void* stream) {
    return train_dw_impl(n_tiles, rows, dacts, g_row0, n_out, acts, x_row0, n_in, x_relu, splits, 0, 0, partial, stream);
}

extern "C" int nefes_train_dw_bias(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts,
                                   int x_row0, int n_in, int x_relu, int splits, int64_t split_stride, float* partial, void* stream) {
    if (split_stride != 0 && split_stride < (int64_t)n_out * (n_in + 1)) return NEFES_E_BADARG;
    return train_dw_impl(n_tiles, rows, dacts, g_row0, n_out, acts, x_row0, n_in, x_relu, splits, 1, split_stride, partial, stream);
}
