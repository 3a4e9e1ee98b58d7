// FusionNet's convolutions (script/models/nerfh_nff.py:356-418: Conv2d(3+C, 64, 3), Conv2d(64, 64, 3) x 2, Conv2d(64, C, 5); stride 1,
// "same" padding) on the rendered feature image of the refinement loop (script/dm/DFM_APR_refine.py:112-120: run_fusion_net on the
// 60x80 render, every iteration, forward and backward to the render), as implicit GEMMs on v_mfma_f32_32x32x2_f32:
//     y[co][p] = bias[co] + sum_{tap, ci} W[co][ci][tap] x[ci][p + tap - pad]        (optionally ReLU)
// M = 32 output channels (A operand: packed weights Wp[ci][tap][co], one dword per lane), N = 32 consecutive pixels (B operand: the
// input plane at the tap's shift, one dword per lane, zero outside the image), K = taps x input channels, two per MFMA.  The four
// waves of a workgroup split the input channels and add their tiles through LDS.  The gradient w.r.t. the input is the same
// kernel on the flipped, transposed weights (packed by the caller), with the ReLU derivative applied to its input on the way in
// (`mask`: the forward output of that layer).  fp32 products, fp32 accumulation: what torch's fp32 convolution computes, in another
// summation order.  The image is a few MB: every operand comes from L2; the kernel is bound by its MFMAs (MIOpen's heuristics
// pick kernels for this one-image shape that take 0.43 ms per iteration for 3.4 GFLOP; batched over eight images they are fine).
#include <hip/hip_runtime.h>

#include "../../include/nefes_hip.h"
#include "layout.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;      // [B][Cin][H][W]
    const float* mask;   // [B][Cin][H][W] or null: x is taken as zero where mask <= 0
    const float* wp;     // [cin_pad][KS*KS][cout_pad] (cin_pad even, cout_pad multiple of 32, zero padded)
    const float* bias;   // [Cout] or null
    float* y;            // [B][Cout][H][W]
    int B, Cin, Cout, H, W, relu, cin_pad, cout_pad;
};

// One k-step = one tap of one pair of input channels (lane half kh holds channel 2s + kh).  What keeps the matrix pipe fed is the
// instruction count per 64-cycle MFMA, so everything a load needs is hoisted out of the K loop:
//   * the byte offset of every tap is a per-lane constant (own pixel where the tap falls outside the image, so that every load
//     has a valid address), relative to the plane of channel 2s: the loads are `global_load_dword v, v_off, s[base]` with a
//     wave-uniform base that advances by two planes per channel pair -- no vector address arithmetic in the loop;
//   * zero padding is one AND of the loaded value with a per-tap, per-lane word (`vm`);
//   * the ReLU derivative of the gradient launches (HAS_MASK) is a compare + select on the value loaded from `mask` at the same offset.
// A stage = one tap ROW of a channel pair (KS k-steps).  Register sets for two channel pairs x KS rows: the loads of pair s + 2,
// row ty are issued as soon as the MFMAs of pair s, row ty have consumed the set, i.e. 2 KS - 1 stages ahead (every operand
// comes from L2; with about one wave per SIMD that latency is covered by the loads in flight, not by other waves).  The K order
// (channel pair, tap row, tap column) and the split of the channel pairs over the four waves are those of the first version of
// this kernel: results are bit-identical to it.
template <int KS, bool HAS_MASK>
__global__ __launch_bounds__(256) void conv2d_kernel(ConvArgs a) {
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W, px_tiles = (HW + 31) / 32, co_tiles = a.cout_pad / 32;
    int bid = blockIdx.x;
    const int ptile = bid % px_tiles;
    bid /= px_tiles;
    const int cot = bid % co_tiles, b = bid / co_tiles;
    const int p = ptile * 32 + n;
    const bool pin = p < HW;
    const int py = pin ? p / a.W : 0, px = pin ? p - py * a.W : 0;
    const int pairs = a.cin_pad / 2;
    const int s_lo = pairs * wave / 4, s_hi = pairs * (wave + 1) / 4;          // this wave's share of the input-channel pairs
    const char* xb = (const char*)(a.x + (size_t)b * a.Cin * HW);
    const char* mb = HAS_MASK ? (const char*)(a.mask + (size_t)b * a.Cin * HW) : nullptr;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int pa = pin ? p : HW - 1;                         // lanes past the last pixel address the last one (their loads are unused)
    uint32_t vm[TAPS], boff[TAPS];      // vm: all ones where the tap lies inside the image (zero padding = one AND per k-step)
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int iy = py + tap / KS - PAD, ix = px + tap % KS - PAD;
        const bool use = pin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        vm[tap] = use ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(vm[tap]));      // a vector register, not a lane mask in two scalar ones: 25 of those spill at 5x5
        boff[tap] = (uint32_t)(kh * HW + (use ? iy * a.W + ix : pa)) * 4u;
    }
    // an odd channel count: the upper lane half of the last pair has no channel (its weights are zero rows of the packed array);
    // it reads the pair's first channel instead of the plane behind the image
    const uint32_t tail_sub = (a.Cin & 1) ? (uint32_t)(kh * HW) * 4u : 0u;
    const uint32_t woff = (uint32_t)(kh * TAPS * a.cout_pad + cot * 32 + n) * 4u;     // + ((2 s) TAPS + tap) cout_pad floats
    const size_t plane2 = (size_t)2 * HW * 4, wpair = (size_t)2 * TAPS * a.cout_pad * 4, wtap = (size_t)a.cout_pad * 4;

    float xv[2][KS][KS], wv[2][KS][KS], mv[2][KS][KS];
    // xs / ms / wrow: wave-uniform bases (plane of channel 2s; first weight of the tap row), every load = base + a per-lane offset
    auto load_row = [&](const char* xs, const char* ms, const char*& wrow, uint32_t sub, int ty, float (&xr)[KS], float (&wr)[KS],
                        float (&mr)[KS]) __attribute__((always_inline)) {
        uint32_t wo = woff;
        asm volatile("" : "+v"(wo));            // keeps the zero-extension next to the loads: base in SGPRs + 32-bit lane offset
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
            const uint32_t o = boff[ty * KS + tx] - sub;
            xr[tx] = *(const float*)(xs + o);
            if (HAS_MASK) mr[tx] = *(const float*)(ms + o);
            wr[tx] = *(const float*)(wrow + wo);
            wrow += wtap;
        }
    };
    auto mma_row = [&](int ty, const float (&xr)[KS], const float (&wr)[KS], const float (&mr)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
            uint32_t v = __float_as_uint(xr[tx]) & vm[ty * KS + tx];
            if (HAS_MASK) v = mr[tx] > 0.f ? v : 0u;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[tx], __uint_as_float(v), acc, 0, 0, 0);
        }
    };
    // bases of channel pair sn (wave-uniform): plane of channel 2 sn, first weight of the pair
    auto pair_bases = [&](int sn, const char*& xs, const char*& ms, const char*& wrow, uint32_t& sub) __attribute__((always_inline)) {
        const size_t so = (size_t)sn * plane2;
        xs = xb + so;
        ms = HAS_MASK ? mb + so : nullptr;
        wrow = (const char*)a.wp + (size_t)sn * wpair;
        sub = sn == pairs - 1 ? tail_sub : 0u;
    };
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
        if (s_lo + pr < s_hi) {                                                  // wave-uniform
            const char *xs, *ms, *wrow;
            uint32_t sub;
            pair_bases(s_lo + pr, xs, ms, wrow, sub);
#pragma unroll
            for (int ty = 0; ty < KS; ++ty) load_row(xs, ms, wrow, sub, ty, xv[pr][ty], wv[pr][ty], mv[pr][ty]);
        }
    for (int s = s_lo; s < s_hi; s += 2) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            if (s + pr < s_hi) {                                                 // wave-uniform
                const bool nxt = s + pr + 2 < s_hi;
                const char *xs, *ms, *wrow;
                uint32_t sub;
                pair_bases(nxt ? s + pr + 2 : s + pr, xs, ms, wrow, sub);
#pragma unroll
                for (int ty = 0; ty < KS; ++ty) {
                    mma_row(ty, xv[pr][ty], wv[pr][ty], mv[pr][ty]);
                    if (nxt) load_row(xs, ms, wrow, sub, ty, xv[pr][ty], wv[pr][ty], mv[pr][ty]);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && pin) {
        float* yb = a.y + (size_t)b * a.Cout * HW + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cot * 32 + nefes_rho(kh, r);
            if (co < a.Cout) {
                float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];     // fixed order: deterministic
                if (a.bias) v += a.bias[co];
                if (a.relu) v = fmaxf(v, 0.f);
                yb[(size_t)co * HW] = v;
            }
        }
    }
}

}  // namespace

extern "C" int nefes_conv2d_same(int B, int Cin, int Cout, int H, int W, int ksize, const float* x, const float* mask,
                                 const float* w_packed, const float* bias, int relu, float* y, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !x || !w_packed || !y) return NEFES_E_BADARG;
    if (ksize != 3 && ksize != 5) return NEFES_E_UNSUPPORTED;
    ConvArgs a;
    a.x = x; a.mask = mask; a.wp = w_packed; a.bias = bias; a.y = y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.relu = relu;
    a.cin_pad = (Cin + 1) / 2 * 2;
    a.cout_pad = (Cout + 31) / 32 * 32;
    const long long blocks = (long long)B * (a.cout_pad / 32) * (((long long)H * W + 31) / 32);
    if (blocks > 0x7fffffffll) return NEFES_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3 && !mask) hipLaunchKernelGGL((conv2d_kernel<3, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (ksize == 3) hipLaunchKernelGGL((conv2d_kernel<3, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (!mask) hipLaunchKernelGGL((conv2d_kernel<5, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv2d_kernel<5, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
