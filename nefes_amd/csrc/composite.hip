// Alpha compositing of RGB + C-channel feature maps and its backward: one wavefront per ray.
// Restates raw2outputs_NeRFH_NFF (script/models/nerfh_nff.py:25-166), variants A/B/C/D of SURVEY.md §8 a9.
//
// Roofline: HBM-bound.  Algorithmic bytes per ray: forward S*R*4 + S*4 read + (3+C+3)*4 write
// (20.1 KB at S=192, R=25); backward 2*S*R*4 + small (38.5 KB).  raw_t is channel-major per ray
// ([N][R][S]) so that lane <-> sample gives 256-byte coalesced rows per channel.
//
// Numerics follow the CPU reference: transmittance = exclusive cumprod accumulated in f64 and rounded
// to f32 per element (torch CPU cumprod semantics); per-ray sums accumulate in f64.
// The backward is division-free (exact for alpha == 1 saturation; SURVEY.md §7 hard part 5).
#include "../../include/nefes_hip.h"
#include "wave.h"

#ifndef NEFES_COMP_STORE
#define NEFES_COMP_STORE 1     /* how composite_bwd4_kernel stores a row: 1 = one 16-byte store; 0 / 2: A/B builds (see there) */
#endif

struct CompArgs {
    int N, S, C, R;
    uint32_t flags;
    float beta_min;
    const float* raw_t;
    const float* z;
    // forward outputs (nullable)
    float *rgb, *feat, *disp, *acc, *depth, *weights, *beta;
    // backward inputs (nullable) and output
    const float *g_rgb, *g_feat, *g_disp, *g_acc, *g_depth, *g_weights, *g_beta;
    float* g_raw_t;
};

// torch.max(1e-10, x) propagates NaN
__device__ __forceinline__ float max_nan(float lo, float x) { return (x != x) ? x : (x > lo ? x : lo); }

template <int Q>
struct RayState {
    float zz[Q], dl[Q], a_c[Q], a_s[Q], a_t[Q], T[Q], Ts[Q], e_c[Q], e_s[Q], e_t[Q];
    bool in[Q];
};

template <int Q>
__device__ __forceinline__ void ray_forward(const CompArgs& p, const float* raw, const float* zr, int lane, RayState<Q>& r) {
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    const int S = p.S, C3 = 3 + p.C;
    const int chS = sigma_only ? 0 : C3, chT = C3 + 4;
    double carry = 1.0, carry_s = 1.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int s = lane + 64 * q;
        r.in[q] = s < S;
        const int sc = r.in[q] ? s : S - 1;
        const float z0 = zr[sc];
        const float z1 = zr[sc + 1 < S ? sc + 1 : sc];
        r.zz[q] = z0;
        r.dl[q] = (sc == S - 1) ? 1e2f : (z1 - z0);                     // :55-60, last delta = 1e2, no |d| scaling
        const float ss = raw[(size_t)chS * S + sc];
        const float st = transient ? raw[(size_t)chT * S + sc] : 0.f;
        r.e_c[q] = expf(-(r.dl[q] * (ss + st)));
        r.a_c[q] = r.in[q] ? 1.f - r.e_c[q] : 0.f;                      // :62-68
        if (transient) {
            r.e_s[q] = expf(-(r.dl[q] * ss));
            r.e_t[q] = expf(-(r.dl[q] * st));
            r.a_s[q] = r.in[q] ? 1.f - r.e_s[q] : 0.f;
            r.a_t[q] = r.in[q] ? 1.f - r.e_t[q] : 0.f;
        } else {
            r.e_s[q] = r.e_c[q]; r.e_t[q] = 1.f; r.a_s[q] = r.a_c[q]; r.a_t[q] = 0.f;
        }
        // exclusive cumprod of (1 - alpha), f64 accumulate, rounded to f32 per element (:71-72)
        const double om = r.in[q] ? (double)(1.f - r.a_c[q]) : 1.0;
        const double inc = wave_incl_prod(om, lane);
        const double prev = __shfl_up(inc, 1);
        r.T[q] = (float)(carry * (lane == 0 ? 1.0 : prev));
        carry *= lane_bcast(inc, 63);
        if (static_only) {                                              // variant B second chain (:95-98)
            const double oms = r.in[q] ? (double)(1.f - r.a_s[q]) : 1.0;
            const double incs = wave_incl_prod(oms, lane);
            const double prevs = __shfl_up(incs, 1);
            r.Ts[q] = (float)(carry_s * (lane == 0 ? 1.0 : prevs));
            carry_s *= lane_bcast(incs, 63);
        } else {
            r.Ts[q] = r.T[q];
        }
    }
}

template <int Q>
__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs p) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= p.N) return;
    const int S = p.S, C = p.C, C3 = 3 + C;
    const float* raw = p.raw_t + (size_t)ray * p.R * S;
    const float* zr = p.z + (size_t)ray * S;
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    // blockIdx.y splits the FEATURE channels of a ray over several workgroups (small frames with many channels: the 80x60
    // refinement frame has 4800 rays x 128 channels, i.e. five waves per SIMD walking 131 reductions each); split 0 also writes
    // everything else.  Every split recomputes the ray's weights (two scans: small next to its share of the channels).
    const int split = blockIdx.y, n_split = gridDim.y;
    const bool lead = split == 0;
    RayState<Q> r;
    ray_forward<Q>(p, raw, zr, lane, r);

    float w[Q], ws[Q], wt[Q], wo[Q];   // combined, static, transient, and the "reported" weights
    double s_acc = 0, s_dep = 0, s_wo = 0, s_beta = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        w[q] = r.a_c[q] * r.T[q];
        ws[q] = static_only ? r.a_s[q] * r.Ts[q] : (transient ? r.a_s[q] * r.T[q] : w[q]);
        wt[q] = (transient && !static_only) ? r.a_t[q] * r.T[q] : 0.f;
        wo[q] = static_only ? ws[q] : w[q];
        if (r.in[q]) {
            s_acc += w[q];
            s_wo += wo[q];
            s_dep += (double)(wo[q] * r.zz[q]);
            if (p.weights && lead) p.weights[(size_t)ray * S + lane + 64 * q] = wo[q];
        }
    }
    const float acc = (float)wave_sum(s_acc);
    if (lane == 0 && p.acc && lead) p.acc[ray] = acc;
    if (sigma_only) return;                                              // variant D: weights + acc only (:83-89)
    if (lead) {
    const float depth = (float)wave_sum(s_dep);
    const float sum_wo = static_only ? (float)wave_sum(s_wo) : acc;
    if (lane == 0) {
        if (p.depth) p.depth[ray] = depth;
        if (p.disp) p.disp[ray] = 1.f / max_nan(1e-10f, depth / sum_wo);   // :115,165
    }
    // rgb = sum w_s * c_s (+ sum w_t * c_t) (:119-131,149 / :101-105 / :153-154)
    float rgb_keep = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = 0, b = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (r.in[q]) {
                const int s = lane + 64 * q;
                a += (double)(ws[q] * raw[(size_t)c * S + s]);
                if (transient && !static_only) b += (double)(wt[q] * raw[(size_t)(C3 + 1 + c) * S + s]);
            }
        float v = (float)wave_sum(a);
        if ((p.flags & NEFES_COMP_WHITE_BKGD) && transient && !static_only) v = v + (1.f - acc);
        if (transient && !static_only) v = v + (float)wave_sum(b);
        if (lane == c) rgb_keep = v;
    }
    if (lane < 3 && p.rgb) p.rgb[(size_t)ray * 3 + lane] = rgb_keep;
    }   // lead
    // features use the same (static) weights, detached (:108-111,122-125,155-157)
    if (p.feat) {
        const int per = ((C + n_split - 1) / n_split + 31) / 32 * 32;      // whole 32-channel groups per split
        const int c_lo = split * per, c_hi = (c_lo + per) < C ? (c_lo + per) : C;
        for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
            float keep = 0.f;
            const int nc = (c_hi - c0) < 64 ? (c_hi - c0) : 64;
            for (int c = 0; c < nc; ++c) {
                double a = 0;
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (r.in[q]) a += (double)(ws[q] * raw[(size_t)(3 + c0 + c) * S + lane + 64 * q]);
                const float v = (float)wave_sum(a);
                if (lane == c) keep = v;
            }
            if (lane < nc) p.feat[(size_t)ray * C + c0 + lane] = keep;
        }
    }
    if (p.beta && lead) {
        float bv = 0.f;
        if (transient && !static_only) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (r.in[q]) s_beta += (double)(wt[q] * raw[(size_t)(C3 + 5) * S + lane + 64 * q]);
            bv = (float)wave_sum(s_beta) + p.beta_min;                   // :133-137
        }
        if (lane == 0) p.beta[ray] = bv;
    }
}

template <int Q>
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs p) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= p.N) return;
    const int S = p.S, C = p.C, C3 = 3 + C;
    const float* raw = p.raw_t + (size_t)ray * p.R * S;
    float* graw = p.g_raw_t + (size_t)ray * p.R * S;
    const float* zr = p.z + (size_t)ray * S;
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    const bool both = transient && !static_only;   // variant A
    RayState<Q> r;
    ray_forward<Q>(p, raw, zr, lane, r);

    float w[Q], ws[Q], wt[Q], wo[Q];
    double s_acc = 0, s_dep = 0, s_wo = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        w[q] = r.a_c[q] * r.T[q];
        ws[q] = static_only ? r.a_s[q] * r.Ts[q] : (transient ? r.a_s[q] * r.T[q] : w[q]);
        wt[q] = both ? r.a_t[q] * r.T[q] : 0.f;
        wo[q] = static_only ? ws[q] : w[q];
        if (r.in[q]) { s_acc += w[q]; s_wo += wo[q]; s_dep += (double)(wo[q] * r.zz[q]); }
    }
    const float acc = (float)wave_sum(s_acc);
    const float depth = (float)wave_sum(s_dep);
    const float sum_wo = static_only ? (float)wave_sum(s_wo) : acc;

    // upstream scalars
    float g_rgb[3] = {0.f, 0.f, 0.f};
    if (p.g_rgb && !sigma_only)
#pragma unroll
        for (int c = 0; c < 3; ++c) g_rgb[c] = p.g_rgb[(size_t)ray * 3 + c];
    float g_acc = p.g_acc ? p.g_acc[ray] : 0.f;
    float g_dep = (p.g_depth && !sigma_only) ? p.g_depth[ray] : 0.f;
    const float g_beta = (p.g_beta && both) ? p.g_beta[ray] : 0.f;
    float g_sumwo = 0.f;   // gradient w.r.t. the disparity denominator
    if (p.g_disp && !sigma_only) {
        const float gd = p.g_disp[ray];
        const float ratio = depth / sum_wo;
        if (ratio > 1e-10f || ratio != ratio) {           // torch.max passes the gradient to the larger operand
            const float disp = 1.f / ratio;
            const float g_ratio = -gd * disp * disp;
            g_dep += g_ratio / sum_wo;
            g_sumwo += -g_ratio * ratio / sum_wo;
        }
    }
    if ((p.flags & NEFES_COMP_WHITE_BKGD) && both) g_acc -= (g_rgb[0] + g_rgb[1] + g_rgb[2]);
    if (!static_only) { g_acc += g_sumwo; g_sumwo = 0.f; }   // same tensor (sum of combined weights) in A/C

    // per-sample gradient w.r.t. the weights, then w.r.t. the transmittance
    float Gs[Q], Gt[Q], Gc[Q], G1[Q];
    double X[Q], X1[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int s = lane + 64 * q;
        Gs[q] = Gt[q] = Gc[q] = G1[q] = 0.f;
        if (r.in[q]) {
            const float gw = p.g_weights ? p.g_weights[(size_t)ray * S + s] : 0.f;
            float col_s = 0.f, col_t = 0.f;
            if (!sigma_only) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    col_s += g_rgb[c] * raw[(size_t)c * S + s];
                    if (both) col_t += g_rgb[c] * raw[(size_t)(C3 + 1 + c) * S + s];
                }
            }
            if (both) {                     // variant A
                Gs[q] = col_s;
                Gt[q] = col_t + g_beta * raw[(size_t)(C3 + 5) * S + s];
                Gc[q] = g_acc + g_dep * r.zz[q] + gw;
            } else if (static_only) {       // variant B
                G1[q] = col_s + g_dep * r.zz[q] + g_sumwo + gw;
                Gc[q] = g_acc;
            } else {                        // variants C / D
                Gc[q] = col_s + g_acc + g_dep * r.zz[q] + gw;
            }
        }
    }
    // X_i = A_i + (1-alpha_i) X_{i+1}: reverse affine scan; B_i = X_{i+1}
    double carry = 0.0, carry1 = 0.0;
    double Bc[Q], B1[Q];
#pragma unroll
    for (int q = Q - 1; q >= 0; --q) {
        {
            double m = r.in[q] ? (double)(1.f - r.a_c[q]) : 1.0;
            double a = r.in[q] ? (double)Gs[q] * r.a_s[q] * (both ? 1.0 : 0.0) + (double)Gt[q] * r.a_t[q] + (double)Gc[q] * r.a_c[q] : 0.0;
            wave_rev_affine(m, a, lane);
            X[q] = a + m * carry;
            const double nxt = __shfl_down(X[q], 1);
            Bc[q] = (lane < 63) ? nxt : carry;
            carry = lane_bcast(X[q], 0);
        }
        if (static_only) {
            double m = r.in[q] ? (double)(1.f - r.a_s[q]) : 1.0;
            double a = r.in[q] ? (double)G1[q] * r.a_s[q] : 0.0;
            wave_rev_affine(m, a, lane);
            X1[q] = a + m * carry1;
            const double nxt = __shfl_down(X1[q], 1);
            B1[q] = (lane < 63) ? nxt : carry1;
            carry1 = lane_bcast(X1[q], 0);
        } else {
            B1[q] = 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (!r.in[q]) continue;
        const int s = lane + 64 * q;
        const double T = r.T[q], Ts = r.Ts[q];
        const double d_ac = (double)Gc[q] * T - T * Bc[q];                       // d L / d alpha (combined)
        double d_as = 0.0, d_at = 0.0;
        if (both) { d_as = (double)Gs[q] * T; d_at = (double)Gt[q] * T; }
        if (static_only) d_as = (double)G1[q] * Ts - Ts * B1[q];
        const double dl = r.dl[q];
        const float g_ss = (float)(transient ? d_as * dl * r.e_s[q] + d_ac * dl * r.e_c[q] : d_ac * dl * r.e_c[q]);
        if (sigma_only) { graw[s] = g_ss; continue; }
        graw[(size_t)C3 * S + s] = g_ss;
#pragma unroll
        for (int c = 0; c < 3; ++c) graw[(size_t)c * S + s] = ws[q] * g_rgb[c];
        if (p.flags & NEFES_COMP_FEAT_WEIGHTS_ONLY) {
            graw[(size_t)3 * S + s] = ws[q];                          // the static weight itself rides in the first feature channel
        } else {
        for (int c = 0; c < C; ++c)
            graw[(size_t)(3 + c) * S + s] = p.g_feat ? ws[q] * p.g_feat[(size_t)ray * C + c] : 0.f;
        }
        if (transient) {
#pragma unroll
            for (int c = 0; c < 3; ++c) graw[(size_t)(C3 + 1 + c) * S + s] = wt[q] * g_rgb[c];
            graw[(size_t)(C3 + 4) * S + s] = (float)(d_at * dl * r.e_t[q] + d_ac * dl * r.e_c[q]);
            graw[(size_t)(C3 + 5) * S + s] = wt[q] * g_beta;
        }
    }
}


// =====================================================================================================================
// Four samples per lane (S a multiple of 64): the kernels above move 4 bytes per lane and load instruction and walk a ray's
// samples in S/64 passes of scans; here a lane owns FOUR CONSECUTIVE samples, so every row of a ray (768 B at S = 192) is one
// 16-byte load per lane, the transmittance scan is four in-lane products plus ONE cross-lane scan, and a ray occupies RW = S/64
// rows of 16 lanes -- 4 / RW rays share a wave (four coarse rays of 64 samples, two 128-sample rays, one ray of 192 or 256).
// Same arithmetic as above: f64 running products / sums rounded to f32 per element, division-free backward.
// =====================================================================================================================
// (Seg<RW>, seg_sum, seg_incl_prod, seg_rev_affine, ld4, el: wave.h -- shared with the fused coarse-pass sampler of sample_pdf.hip)
struct Ray4 {
    float zz[4], dl[4], a_c[4], a_s[4], a_t[4], T[4], Ts[4], e_c[4], e_s[4], e_t[4];
};
template <int RW>
__device__ __forceinline__ void ray_forward4(const CompArgs& p, const float* raw, const float* zr, int sl, Ray4& r) {
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    const int S = p.S, C3 = 3 + p.C;
    const int chS = sigma_only ? 0 : C3, chT = C3 + 4;
    const int s0 = 4 * sl;
    const float4 z4 = ld4(zr + s0);
    const float4 ss4 = ld4(raw + (size_t)chS * S + s0);
    float4 st4 = {0.f, 0.f, 0.f, 0.f};
    if (transient) st4 = ld4(raw + (size_t)chT * S + s0);
    const float z_next = __shfl_down(z4.x, 1);                     // first depth of the next lane's group
    double om[4], oms[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float z0 = el(z4, k);
        const float z1 = k < 3 ? el(z4, k + 1) : z_next;
        r.zz[k] = z0;
        r.dl[k] = (s0 + k == S - 1) ? 1e2f : (z1 - z0);
        const float ss = el(ss4, k), st = el(st4, k);
        r.e_c[k] = expf(-(r.dl[k] * (ss + st)));
        r.a_c[k] = 1.f - r.e_c[k];
        if (transient) {
            r.e_s[k] = expf(-(r.dl[k] * ss));
            r.e_t[k] = expf(-(r.dl[k] * st));
            r.a_s[k] = 1.f - r.e_s[k];
            r.a_t[k] = 1.f - r.e_t[k];
        } else {
            r.e_s[k] = r.e_c[k]; r.e_t[k] = 1.f; r.a_s[k] = r.a_c[k]; r.a_t[k] = 0.f;
        }
        om[k] = (double)(1.f - r.a_c[k]);
        oms[k] = (double)(1.f - r.a_s[k]);
    }
    {   // exclusive cumprod: the lane's own running product, times the product of everything in front of the lane
        const double p0 = om[0], p1 = p0 * om[1], p2 = p1 * om[2], p3 = p2 * om[3];
        const double inc = seg_incl_prod<RW>(p3, sl);
        const double prev = __shfl_up(inc, 1);
        const double E = sl == 0 ? 1.0 : prev;
        r.T[0] = (float)E; r.T[1] = (float)(E * p0); r.T[2] = (float)(E * p1); r.T[3] = (float)(E * p2);
    }
    if (static_only) {
        const double p0 = oms[0], p1 = p0 * oms[1], p2 = p1 * oms[2], p3 = p2 * oms[3];
        const double inc = seg_incl_prod<RW>(p3, sl);
        const double prev = __shfl_up(inc, 1);
        const double E = sl == 0 ? 1.0 : prev;
        r.Ts[0] = (float)E; r.Ts[1] = (float)(E * p0); r.Ts[2] = (float)(E * p1); r.Ts[3] = (float)(E * p2);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.Ts[k] = r.T[k];
    }
}

template <int RW>
__global__ __launch_bounds__(256) void composite_fwd4_kernel(CompArgs p) {
    constexpr int LPR = Seg<RW>::LPR, RPW = Seg<RW>::RPW;
    const int lane = threadIdx.x & 63;
    const int seg = lane / LPR, sl = lane - seg * LPR;
    const int ray_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + seg;
    const bool live = seg < RPW && ray_raw < p.N;
    const int ray = live ? ray_raw : p.N - 1;            // idle lanes shadow the last ray (loads stay in bounds, nothing is stored)
    const int S = p.S, C = p.C, C3 = 3 + C;
    const float* raw = p.raw_t + (size_t)ray * p.R * S;
    const float* zr = p.z + (size_t)ray * S;
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    const bool both = transient && !static_only;
    const int split = blockIdx.y, n_split = gridDim.y;
    const bool lead = split == 0;
    const int s0 = 4 * sl;
    Ray4 r;
    ray_forward4<RW>(p, raw, zr, sl, r);

    float w[4], ws[4], wt[4], wo[4];
    double s_acc = 0, s_dep = 0, s_wo = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = r.a_c[k] * r.T[k];
        ws[k] = static_only ? r.a_s[k] * r.Ts[k] : (transient ? r.a_s[k] * r.T[k] : w[k]);
        wt[k] = both ? r.a_t[k] * r.T[k] : 0.f;
        wo[k] = static_only ? ws[k] : w[k];
        s_acc += w[k];
        s_wo += wo[k];
        s_dep += (double)(wo[k] * r.zz[k]);
    }
    if (p.weights && lead && live) *(float4*)(p.weights + (size_t)ray * S + s0) = make_float4(wo[0], wo[1], wo[2], wo[3]);
    const float acc = (float)seg_sum<RW>(s_acc, lane);
    if (sl == 0 && p.acc && lead && live) p.acc[ray] = acc;
    if (sigma_only) return;
    if (lead) {
        const float depth = (float)seg_sum<RW>(s_dep, lane);
        const float sum_wo = static_only ? (float)seg_sum<RW>(s_wo, lane) : acc;
        if (sl == 0 && live) {
            if (p.depth) p.depth[ray] = depth;
            if (p.disp) p.disp[ray] = 1.f / max_nan(1e-10f, depth / sum_wo);
        }
        float4 c4[3], t4[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            c4[c] = ld4(raw + (size_t)c * S + s0);
            if (both) t4[c] = ld4(raw + (size_t)(C3 + 1 + c) * S + s0);
        }
        float rgb_keep = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double a = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a += (double)(ws[k] * el(c4[c], k));
                if (both) b += (double)(wt[k] * el(t4[c], k));
            }
            float v = (float)seg_sum<RW>(a, lane);
            if ((p.flags & NEFES_COMP_WHITE_BKGD) && both) v = v + (1.f - acc);
            if (both) v = v + (float)seg_sum<RW>(b, lane);
            if (sl == c) rgb_keep = v;
        }
        if (sl < 3 && p.rgb && live) p.rgb[(size_t)ray * 3 + sl] = rgb_keep;
    }
    if (p.feat) {
        const int per = ((C + n_split - 1) / n_split + 31) / 32 * 32;
        const int c_lo = split * per, c_hi = (c_lo + per) < C ? (c_lo + per) : C;
        for (int c0 = c_lo; c0 < c_hi; c0 += LPR) {
            float keep = 0.f;
            const int nc = (c_hi - c0) < LPR ? (c_hi - c0) : LPR;
            for (int cb = 0; cb < nc; cb += 4) {                         // four rows requested before the first is consumed
                float4 f4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cb + i < nc ? cb + i : nc - 1;
                    f4[i] = ld4(raw + (size_t)(3 + c0 + c) * S + s0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double a = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) a += (double)(ws[k] * el(f4[i], k));
                    const float v = (float)seg_sum<RW>(a, lane);
                    if (sl == cb + i) keep = v;
                }
            }
            if (sl < nc && live) p.feat[(size_t)ray * C + c0 + sl] = keep;
        }
    }
    if (p.beta && lead) {
        float bv = 0.f;
        if (both) {
            const float4 b4 = ld4(raw + (size_t)(C3 + 5) * S + s0);
            double sb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sb += (double)(wt[k] * el(b4, k));
            bv = (float)seg_sum<RW>(sb, lane) + p.beta_min;
        }
        if (sl == 0 && live) p.beta[ray] = bv;
    }
}

template <int RW>
__global__ __launch_bounds__(256) void composite_bwd4_kernel(CompArgs p) {
    constexpr int LPR = Seg<RW>::LPR, RPW = Seg<RW>::RPW;
    const int lane = threadIdx.x & 63;
    const int seg = lane / LPR, sl = lane - seg * LPR;
    const int ray_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + seg;
    const bool live = seg < RPW && ray_raw < p.N;
    const int ray = live ? ray_raw : p.N - 1;
    const int S = p.S, C = p.C, C3 = 3 + C;
    const float* raw = p.raw_t + (size_t)ray * p.R * S;
    float* graw = p.g_raw_t + (size_t)ray * p.R * S;
    const float* zr = p.z + (size_t)ray * S;
    const bool transient = p.flags & NEFES_COMP_TRANSIENT, sigma_only = p.flags & NEFES_COMP_SIGMA_ONLY;
    const bool static_only = p.flags & NEFES_COMP_STATIC_ONLY;
    const bool both = transient && !static_only;
    const int s0 = 4 * sl;
    // rows the per-sample weight gradients need, requested before the scans
    float4 c4[3], t4[3], b4 = {0.f, 0.f, 0.f, 0.f}, gw4 = {0.f, 0.f, 0.f, 0.f};
    if (!sigma_only) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            c4[c] = ld4(raw + (size_t)c * S + s0);
            if (both) t4[c] = ld4(raw + (size_t)(C3 + 1 + c) * S + s0);
        }
        if (both) b4 = ld4(raw + (size_t)(C3 + 5) * S + s0);
    }
    if (p.g_weights) gw4 = ld4(p.g_weights + (size_t)ray * S + s0);
    Ray4 r;
    ray_forward4<RW>(p, raw, zr, sl, r);

    float w[4], ws[4], wt[4], wo[4];
    double s_acc = 0, s_dep = 0, s_wo = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = r.a_c[k] * r.T[k];
        ws[k] = static_only ? r.a_s[k] * r.Ts[k] : (transient ? r.a_s[k] * r.T[k] : w[k]);
        wt[k] = both ? r.a_t[k] * r.T[k] : 0.f;
        wo[k] = static_only ? ws[k] : w[k];
        s_acc += w[k]; s_wo += wo[k]; s_dep += (double)(wo[k] * r.zz[k]);
    }
    const float acc = (float)seg_sum<RW>(s_acc, lane);
    const float depth = (float)seg_sum<RW>(s_dep, lane);
    const float sum_wo = static_only ? (float)seg_sum<RW>(s_wo, lane) : acc;

    float g_rgb[3] = {0.f, 0.f, 0.f};
    if (p.g_rgb && !sigma_only)
#pragma unroll
        for (int c = 0; c < 3; ++c) g_rgb[c] = p.g_rgb[(size_t)ray * 3 + c];
    float g_acc = p.g_acc ? p.g_acc[ray] : 0.f;
    float g_dep = (p.g_depth && !sigma_only) ? p.g_depth[ray] : 0.f;
    const float g_beta = (p.g_beta && both) ? p.g_beta[ray] : 0.f;
    float g_sumwo = 0.f;
    if (p.g_disp && !sigma_only) {
        const float gd = p.g_disp[ray];
        const float ratio = depth / sum_wo;
        if (ratio > 1e-10f || ratio != ratio) {
            const float disp = 1.f / ratio;
            const float g_ratio = -gd * disp * disp;
            g_dep += g_ratio / sum_wo;
            g_sumwo += -g_ratio * ratio / sum_wo;
        }
    }
    if ((p.flags & NEFES_COMP_WHITE_BKGD) && both) g_acc -= (g_rgb[0] + g_rgb[1] + g_rgb[2]);
    if (!static_only) { g_acc += g_sumwo; g_sumwo = 0.f; }

    float Gs[4], Gt[4], Gc[4], G1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Gs[k] = Gt[k] = Gc[k] = G1[k] = 0.f;
        const float gw = el(gw4, k);
        float col_s = 0.f, col_t = 0.f;
        if (!sigma_only) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                col_s += g_rgb[c] * el(c4[c], k);
                if (both) col_t += g_rgb[c] * el(t4[c], k);
            }
        }
        if (both) {
            Gs[k] = col_s;
            Gt[k] = col_t + g_beta * el(b4, k);
            Gc[k] = g_acc + g_dep * r.zz[k] + gw;
        } else if (static_only) {
            G1[k] = col_s + g_dep * r.zz[k] + g_sumwo + gw;
            Gc[k] = g_acc;
        } else {
            Gc[k] = col_s + g_acc + g_dep * r.zz[k] + gw;
        }
    }
    // X_i = A_i + (1 - alpha_i) X_{i+1} over the ray: the lane's four maps composed, one reverse scan across lanes, then the
    // lane's own elements by back-substitution from the next lane's X.  B_i = X_{i+1}.
    double Bc[4], B1[4];
    {
        double mk[4], ak[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk[k] = (double)(1.f - r.a_c[k]);
            ak[k] = (double)Gs[k] * r.a_s[k] * (both ? 1.0 : 0.0) + (double)Gt[k] * r.a_t[k] + (double)Gc[k] * r.a_c[k];
        }
        double m = ((mk[0] * mk[1]) * mk[2]) * mk[3];
        double a = ak[0] + mk[0] * (ak[1] + mk[1] * (ak[2] + mk[2] * ak[3]));
        seg_rev_affine<RW>(m, a, sl);
        const double nxt = __shfl_down(a, 1);
        double X = sl + 1 < LPR ? nxt : 0.0;
#pragma unroll
        for (int k = 3; k >= 0; --k) { Bc[k] = X; X = ak[k] + mk[k] * X; }
    }
    if (static_only) {
        double mk[4], ak[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { mk[k] = (double)(1.f - r.a_s[k]); ak[k] = (double)G1[k] * r.a_s[k]; }
        double m = ((mk[0] * mk[1]) * mk[2]) * mk[3];
        double a = ak[0] + mk[0] * (ak[1] + mk[1] * (ak[2] + mk[2] * ak[3]));
        seg_rev_affine<RW>(m, a, sl);
        const double nxt = __shfl_down(a, 1);
        double X = sl + 1 < LPR ? nxt : 0.0;
#pragma unroll
        for (int k = 3; k >= 0; --k) { B1[k] = X; X = ak[k] + mk[k] * X; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) B1[k] = 0.0;
    }
    if (!live) return;
    float gss[4], gst[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double T = r.T[k], Ts = r.Ts[k];
        const double d_ac = (double)Gc[k] * T - T * Bc[k];
        double d_as = 0.0, d_at = 0.0;
        if (both) { d_as = (double)Gs[k] * T; d_at = (double)Gt[k] * T; }
        if (static_only) d_as = (double)G1[k] * Ts - Ts * B1[k];
        const double dl = r.dl[k];
        gss[k] = (float)(transient ? d_as * dl * r.e_s[k] + d_ac * dl * r.e_c[k] : d_ac * dl * r.e_c[k]);
        gst[k] = (float)(d_at * dl * r.e_t[k] + d_ac * dl * r.e_c[k]);
    }
    // (Round 5: with a field kernel of ANOTHER stream resident on the same CUs this kernel once wrote wrong values -- the last 16 lanes of
    // the green transient row.  Not the store: the `v_pk_mul_f32 ..., op_sel:[0,1]` hipcc's SLP vectoriser had made of `wt[k] * g_rgb[1]`.
    // That instruction form returns a wrong low result on lanes 48..63 next to another queue's v_mfma_f32_32x32x16_f16 kernel
    // (tools/store_hazard.py).  The library is built without the vectoriser now; NEFES_COMP_STORE = 0 / 2 are the store
    // variants tried on the way -- two 8-byte stores happened to make the compiler drop the packed multiply, waiting for every store did not.)
    auto store_row = [&](int ch, const float (&v)[4]) {
        float* q = graw + (size_t)ch * S + s0;
#if NEFES_COMP_STORE == 0
        const float2 lo = make_float2(v[0], v[1]), hi = make_float2(v[2], v[3]);
        asm volatile("global_store_dwordx2 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:8" :: "v"(q), "v"(lo), "v"(hi) : "memory");
#else       // A/B builds (make EXTRA=-DNEFES_COMP_STORE=1|2): the original 16-byte store; the same + a wait for every store's completion
        *(float4*)q = make_float4(v[0], v[1], v[2], v[3]);
#if NEFES_COMP_STORE == 2
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
#endif
    };
    if (sigma_only) { store_row(0, gss); return; }
    store_row(C3, gss);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v[4] = {ws[0] * g_rgb[c], ws[1] * g_rgb[c], ws[2] * g_rgb[c], ws[3] * g_rgb[c]};
        store_row(c, v);
    }
    if (p.flags & NEFES_COMP_FEAT_WEIGHTS_ONLY) {
        // the factored head (field_bwd_h3.hip FH): d loss / d feature channel c at sample s is w_s[s] g_feat[c], which the field backward
        // forms itself from g_feat (one row per RAY) and the static weight -- written HERE, into the first feature channel's row, in
        // place of C rows of products
        const float v[4] = {ws[0], ws[1], ws[2], ws[3]};
        store_row(3, v);
    } else {
    const float* gf = p.g_feat ? p.g_feat + (size_t)ray * C : nullptr;
    for (int c = 0; c < C; ++c) {
        const float gfc = gf ? gf[c] : 0.f;
        const float v[4] = {gf ? ws[0] * gfc : 0.f, gf ? ws[1] * gfc : 0.f, gf ? ws[2] * gfc : 0.f, gf ? ws[3] * gfc : 0.f};
        store_row(3 + c, v);
    }
    }
    if (transient) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v[4] = {wt[0] * g_rgb[c], wt[1] * g_rgb[c], wt[2] * g_rgb[c], wt[3] * g_rgb[c]};
            store_row(C3 + 1 + c, v);
        }
        store_row(C3 + 4, gst);
        const float v[4] = {wt[0] * g_beta, wt[1] * g_beta, wt[2] * g_beta, wt[3] * g_beta};
        store_row(C3 + 5, v);
    }
}

#ifndef NEFES_COMPOSITE_LEGACY
#define NEFES_COMPOSITE_LEGACY 0      /* 1: one sample per lane for every S (A/B builds) */
#endif
#define NEFES_COMP_MAX_S 512           /* one sample per lane in up to eight passes; four per lane (S % 64 == 0) up to 256 */
static bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }
static int comp_args(CompArgs& a, int N, int S, int C, uint32_t flags) {
    if (N <= 0 || S <= 1 || C < 0) return NEFES_E_BADARG;
    if (S > NEFES_COMP_MAX_S) return NEFES_E_UNSUPPORTED;      // (the sampler's bound too: sample_pdf.hip SP_MAX_S)
    a.N = N; a.S = S; a.C = C; a.flags = flags;
    a.R = (flags & NEFES_COMP_SIGMA_ONLY) ? 1 : ((flags & NEFES_COMP_TRANSIENT) ? 3 + C + 6 : 3 + C + 1);
    return 0;
}

extern "C" int nefes_composite_fwd(int N, int S, int C, uint32_t flags, float beta_min, const float* raw_t,
                                   const float* z, float* rgb, float* feat, float* disp, float* acc, float* depth,
                                   float* weights, float* beta, void* stream) {
    CompArgs a = {};
    int rc = comp_args(a, N, S, C, flags);
    if (rc) return rc;
    if (!raw_t || !z) return NEFES_E_BADARG;
    a.beta_min = beta_min; a.raw_t = raw_t; a.z = z;
    a.rgb = rgb; a.feat = feat; a.disp = disp; a.acc = acc; a.depth = depth; a.weights = weights; a.beta = beta;
    // small frames with many feature channels: split the channels of a ray over up to four workgroups (composite_fwd_kernel)
    const int n_split = (feat && C >= 64 && N < 40000) ? ((C + 31) / 32 < 4 ? (C + 31) / 32 : 4) : 1;
    const dim3 grid((N + 3) / 4, n_split), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (S % 64 == 0 && S <= 256 && !NEFES_COMPOSITE_LEGACY && aligned16(raw_t) && aligned16(z) && aligned16(weights)) {   // four samples per lane
        const int rw = S / 64, rpw = 4 / rw;
        const dim3 grid4((N + 4 * rpw - 1) / (4 * rpw), n_split);
        switch (rw) {
            case 1: hipLaunchKernelGGL(composite_fwd4_kernel<1>, grid4, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL(composite_fwd4_kernel<2>, grid4, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL(composite_fwd4_kernel<3>, grid4, block, 0, st, a); break;
            default: hipLaunchKernelGGL(composite_fwd4_kernel<4>, grid4, block, 0, st, a); break;
        }
        return (int)hipGetLastError();
    }
    switch ((S + 63) / 64) {
        case 1: hipLaunchKernelGGL(composite_fwd_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(composite_fwd_kernel<2>, grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL(composite_fwd_kernel<3>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(composite_fwd_kernel<4>, grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL(composite_fwd_kernel<5>, grid, block, 0, st, a); break;      // 64 + 256 samples
        case 6: hipLaunchKernelGGL(composite_fwd_kernel<6>, grid, block, 0, st, a); break;      // 128 + 256
        case 7: hipLaunchKernelGGL(composite_fwd_kernel<7>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(composite_fwd_kernel<8>, grid, block, 0, st, a); break;
    }
    return (int)hipGetLastError();
}

extern "C" int nefes_composite_bwd(int N, int S, int C, uint32_t flags, const float* raw_t, const float* z,
                                   const float* g_rgb, const float* g_feat, const float* g_disp, const float* g_acc,
                                   const float* g_depth, const float* g_weights, const float* g_beta, float* g_raw_t,
                                   void* stream) {
    CompArgs a = {};
    int rc = comp_args(a, N, S, C, flags);
    if (rc) return rc;
    if (!raw_t || !z || !g_raw_t) return NEFES_E_BADARG;
    a.raw_t = raw_t; a.z = z; a.g_rgb = g_rgb; a.g_feat = g_feat; a.g_disp = g_disp; a.g_acc = g_acc;
    a.g_depth = g_depth; a.g_weights = g_weights; a.g_beta = g_beta; a.g_raw_t = g_raw_t;
    const dim3 grid((N + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (S % 64 == 0 && S <= 256 && !NEFES_COMPOSITE_LEGACY && aligned16(raw_t) && aligned16(z) && aligned16(g_raw_t) && aligned16(g_weights)) {
        const int rw = S / 64, rpw = 4 / rw;
        const dim3 grid4((N + 4 * rpw - 1) / (4 * rpw));
        switch (rw) {
            case 1: hipLaunchKernelGGL(composite_bwd4_kernel<1>, grid4, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL(composite_bwd4_kernel<2>, grid4, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL(composite_bwd4_kernel<3>, grid4, block, 0, st, a); break;
            default: hipLaunchKernelGGL(composite_bwd4_kernel<4>, grid4, block, 0, st, a); break;
        }
        return (int)hipGetLastError();
    }
    switch ((S + 63) / 64) {
        case 1: hipLaunchKernelGGL(composite_bwd_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(composite_bwd_kernel<2>, grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL(composite_bwd_kernel<3>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(composite_bwd_kernel<4>, grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL(composite_bwd_kernel<5>, grid, block, 0, st, a); break;      // 64 + 256 samples
        case 6: hipLaunchKernelGGL(composite_bwd_kernel<6>, grid, block, 0, st, a); break;      // 128 + 256
        case 7: hipLaunchKernelGGL(composite_bwd_kernel<7>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(composite_bwd_kernel<8>, grid, block, 0, st, a); break;
    }
    return (int)hipGetLastError();
}
