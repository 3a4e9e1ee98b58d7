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

template <int KS>
__global__ __launch_bounds__(256) void conv2d_kernel(ConvArgs a) {
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, kh = lane >> 5;
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
    const float* xb = a.x + (size_t)b * a.Cin * HW;
    const float* mb = a.mask ? a.mask + (size_t)b * a.Cin * HW : nullptr;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // K runs channel-pair-major: for a pair of input channels, all taps.  A stage = CPS channel pairs x TAPS k-steps, fully unrolled
    // (no branches inside): its 2 x CPS x TAPS loads are requested while the previous stage's MFMAs issue -- every operand comes from
    // L2 (~1 us), and with one or two waves per SIMD that latency is covered by the loads in flight, not by other waves.  The taps
    // of one channel read neighbouring addresses of one plane; validity of a tap (zero padding) is one bit per lane, computed once.
    constexpr int CPS = KS == 3 ? 2 : 1, NST = CPS * TAPS;      // (twice as many per stage: 119.5 -> 125 ms per 50 iterations)
    uint32_t vbits = 0;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int iy = py + tap / KS - PAD, ix = px + tap % KS - PAD;
        if (pin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) vbits |= 1u << tap;
    }
    const int pa = pin ? p : HW - 1;                         // lanes past the last pixel address the last one (their loads are unused)
    const float* xp = xb + pa - (PAD * a.W + PAD);           // tap (ty, tx) of channel ci: xp[ci * HW + ty * W + tx]
    const float* mp = mb ? mb + pa - (PAD * a.W + PAD) : nullptr;
    const float* wl = a.wp + cot * 32 + n;                   // wl[(ci * TAPS + tap) * cout_pad]
    auto fetch = [&](int s0, float (&av)[NST], float (&bv)[NST]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
            const int s = s0 + c;
            const bool live = s < s_hi;
            const int ci = 2 * (live ? s : s_lo) + kh;
            const bool chan = live && ci < a.Cin;
            const float* xc = xp + (size_t)(chan ? ci : 0) * HW;
            const float* mc = mp ? mp + (size_t)(chan ? ci : 0) * HW : nullptr;
            const float* wc = wl + (size_t)ci * TAPS * a.cout_pad;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const bool use = chan && ((vbits >> tap) & 1u);
                const int o = use ? (tap / KS) * a.W + tap % KS : PAD * a.W + PAD;      // (unused: the lane's own pixel, a valid address)
                float v = xc[o];
                if (mc) v = mc[o] > 0.f ? v : 0.f;
                bv[c * TAPS + tap] = use ? v : 0.f;
                av[c * TAPS + tap] = wc[(size_t)tap * a.cout_pad];
            }
        }
    };
    float a0[NST], b0[NST], a1[NST], b1[NST];
    fetch(s_lo, a0, b0);
    for (int s = s_lo; s < s_hi; s += 2 * CPS) {
        fetch(s + CPS, a1, b1);
#pragma unroll
        for (int u = 0; u < NST; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
        fetch(s + 2 * CPS, a0, b0);
#pragma unroll
        for (int u = 0; u < NST; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);   // (past the end: zeros)
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
    if (ksize == 3) hipLaunchKernelGGL(conv2d_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv2d_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
