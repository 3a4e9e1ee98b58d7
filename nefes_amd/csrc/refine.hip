// Glue of the per-image refinement loop around the render path, as kernels: in torch the
// twelve numbers of the pose chain cost ~100 launches per iteration (forward + autograd), the feature loss on the up-sampled
// 128 x 220 x 300 maps another ~35 elementwise passes over 34 MB each -- a fifth of an 80x60 refinement iteration.
//
//   pose_compose      LearnPose.forward (script/models/poses.py:43-50, lietorch=False: utils/lie_group_helper.py:60-81)
//                     followed by fix_coord_supp (script/dm/direct_pose_model.py:224-231):
//                         c2w = [ Exp(r) R0 | ((t + t0) sc + move) sc2 ]            (3x4, one thread, float64 inside)
//                     and its backward to (r, t): analytic derivative of the Rodrigues formula.
//   cosine_loss       feature_loss (script/dm/DFM_pose_refine.py:211-233, per_pixel=False): 1 - mean_c cos(a_c, b_c) over the
//                     pixels of [C, P] maps, torch.nn.CosineSimilarity(dim=1, eps=1e-6) semantics (each norm clamped at eps),
//                     float64 accumulation; backward d loss / d a in one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/nefes_hip.h"
#include "bicubic.h"
#include "wave.h"

namespace {

struct PoseArgs {
    const float* r;       // [3]
    const float* t;       // [3]
    const float* init;    // [4,4] row-major (rows 0..2 used)
    float sc, sc2, mv[3];
};

__device__ void rodrigues(const double r[3], double R[3][3], double dR[3][3][3]) {
    const double rho = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double n = rho + 1e-15;                                           // lie_group_helper.py:66
    const double K[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
    double K2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
    double a, b, da, db;                                                     // a = sin n / n, b = (1 - cos n) / n^2 and d/dn
    if (n < 1e-4) {
        const double n2 = n * n;
        a = 1 - n2 / 6;
        b = 0.5 - n2 / 24;
        da = -n / 3 + n * n2 / 30;
        db = -n / 12 + n * n2 / 180;
        if (rho == 0) b = 0;            // the reference's fp32 (1 - cos 1e-15) / 1e-30 is exactly 0; K^2 is 0 there anyway
    } else {
        const double s = sin(n), c = cos(n);
        a = s / n;
        b = (1 - c) / (n * n);
        da = (n * c - s) / (n * n);
        db = (n * s - 2 * (1 - c)) / (n * n * n);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = (i == j ? 1.0 : 0.0) + a * K[i][j] + b * K2[i][j];
    if (!dR) return;
    for (int k = 0; k < 3; ++k) {
        const double dn = rho > 0 ? r[k] / rho : 0.0;                         // torch: d|r|/dr = 0 at r = 0
        double E[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};                   // d K / d r_k
        if (k == 0) { E[1][2] = -1; E[2][1] = 1; }
        if (k == 1) { E[0][2] = 1; E[2][0] = -1; }
        if (k == 2) { E[0][1] = -1; E[1][0] = 1; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double ek = 0;                                                // E K + K E
                for (int m = 0; m < 3; ++m) ek += E[i][m] * K[m][j] + K[i][m] * E[m][j];
                dR[k][i][j] = da * dn * K[i][j] + a * E[i][j] + db * dn * K2[i][j] + b * ek;
            }
    }
}

__global__ void pose_compose_fwd_kernel(PoseArgs p, float* __restrict__ c2w) {
    if (threadIdx.x != 0) return;
    p.r += 3 * blockIdx.x; p.t += 3 * blockIdx.x; p.init += 16 * blockIdx.x; c2w += 12 * blockIdx.x;      // one pose per block
    const double r[3] = {p.r[0], p.r[1], p.r[2]};
    double R[3][3];
    rodrigues(r, R, nullptr);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double v = 0;
            for (int m = 0; m < 3; ++m) v += R[i][m] * (double)p.init[m * 4 + j];
            c2w[i * 4 + j] = (float)v;
        }
        float tt = p.t[i] + p.init[i * 4 + 3];                              // poses.py:49, then direct_pose_model.py:227-231 in fp32
        tt *= p.sc;
        tt += p.mv[i];
        tt *= p.sc2;
        c2w[i * 4 + 3] = tt;
    }
}

__global__ void pose_compose_bwd_kernel(PoseArgs p, const float* __restrict__ g, float* __restrict__ g_r, float* __restrict__ g_t) {
    if (threadIdx.x != 0) return;
    p.r += 3 * blockIdx.x; p.t += 3 * blockIdx.x; p.init += 16 * blockIdx.x;
    g += 12 * blockIdx.x; g_r += 3 * blockIdx.x; g_t += 3 * blockIdx.x;
    const double r[3] = {p.r[0], p.r[1], p.r[2]};
    double R[3][3], dR[3][3][3];
    rodrigues(r, R, dR);
    double GR[3][3];                                                          // d L / d R = G[:, :3] R0^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0;
            for (int m = 0; m < 3; ++m) v += (double)g[i * 4 + m] * (double)p.init[j * 4 + m];
            GR[i][j] = v;
        }
    for (int k = 0; k < 3; ++k) {
        double v = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v += GR[i][j] * dR[k][i][j];
        g_r[k] = (float)v;
        g_t[k] = (float)((double)g[k * 4 + 3] * (double)p.sc * (double)p.sc2);
    }
}

// ---- FusionNet's BatchNorm2d in TRAIN mode with frozen affine parameters (nerfh_nff.py:356-418; the reference runs the net with the
// module in train mode, one image at a time: the output is normalised by THAT image's statistics) -------------------------------------
// x [B, C, P] -> y; one workgroup per (group, channel), groups = 1 (statistics over the whole batch: torch's BatchNorm) or B (per image:
// what the reference's one-image-at-a-time loop computes; PoseRefiner(images=B)).  float64 sums in a fixed order; biased variance for the
// output, unbiased for the running estimate (torch.nn.BatchNorm2d semantics; momentum m: r <- (1 - m) r + m s).  save [groups*C][2] doubles
// = (mean, 1 / sqrt(var + eps)).  In-house because the loop's other kernels are: no library launch is left in an iteration, and no
// foreign code shares the CUs with the field kernels of another stream.
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(int B, int C, long P, int groups, const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, double eps, double momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, long long* __restrict__ batches, float* __restrict__ y, double* __restrict__ save) {
    __shared__ double sh[256];
    const int c = blockIdx.x % C, g = blockIdx.x / C;
    const int per = B / groups;                              // images per group
    const long n = (long)per * P;
    // element i of this (group, channel): one image per group (the loop's case) needs no division
    auto at = [&](long i) -> long { return per == 1 ? ((long)g * C + c) * P + i : ((long)(g * per + i / P) * C + c) * P + i % P; };
    auto reduce = [&](double v) -> double {                 // wave sums on DPP (wave.h), the four of them added in wave order
        const double ws_ = wave_sum(v);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ws_;
        __syncthreads();
        const double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        __syncthreads();
        return r;
    };
    const double ww = w ? (double)w[c] : 1.0, bb = b ? (double)b[c] : 0.0;
    double mean, ss, var, inv;
    constexpr int KE = 20;                                   // elements a thread keeps in registers (the loop's 60 x 80 image: 18.75)
    if (per == 1 && n <= 256 * KE) {                         // one read of x: the three passes run on registers
        const float* xp = x + ((long)g * C + c) * P;
        float v[KE];
        double acc = 0;
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const long i = threadIdx.x + 256 * k;
            v[k] = i < n ? xp[i] : 0.f;
            acc += (double)v[k];
        }
        mean = reduce(acc) / (double)n;
        acc = 0;
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const double d = (double)v[k] - mean;
            acc += threadIdx.x + 256 * k < n ? d * d : 0.0;
        }
        ss = reduce(acc); var = ss / (double)n; inv = 1.0 / sqrt(var + eps);
        float* yp = y + ((long)g * C + c) * P;
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const long i = threadIdx.x + 256 * k;
            if (i < n) yp[i] = (float)(((double)v[k] - mean) * inv * ww + bb);
        }
    } else {
        double acc = 0;
        for (long i = threadIdx.x; i < n; i += 256) acc += (double)x[at(i)];
        mean = reduce(acc) / (double)n;
        acc = 0;
        for (long i = threadIdx.x; i < n; i += 256) {
            const double d = (double)x[at(i)] - mean;
            acc += d * d;
        }
        ss = reduce(acc); var = ss / (double)n; inv = 1.0 / sqrt(var + eps);
        for (long i = threadIdx.x; i < n; i += 256) {
            const long o = at(i);
            y[o] = (float)(((double)x[o] - mean) * inv * ww + bb);
        }
    }
    if (threadIdx.x == 0) {
        save[((long)g * C + c) * 2 + 0] = mean;
        save[((long)g * C + c) * 2 + 1] = inv;
        if (blockIdx.x == 0 && batches) *batches += 1;       // num_batches_tracked
        if (groups == 1 && running_mean && running_var) {
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * (n > 1 ? ss / (double)(n - 1) : var));
        }
    }
}

// d x = (w / sigma) (d y - mean(d y) - xhat mean(d y xhat)) over the group's elements
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(int B, int C, long P, int groups, const float* __restrict__ x, const float* __restrict__ w,
                                                           const double* __restrict__ save, const float* __restrict__ gy, float* __restrict__ gx) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x % C, g = blockIdx.x / C;
    const int per = B / groups;
    const long n = (long)per * P;
    const double mean = save[((long)g * C + c) * 2 + 0], inv = save[((long)g * C + c) * 2 + 1];
    auto at = [&](long i) -> long { return per == 1 ? ((long)g * C + c) * P + i : ((long)(g * per + i / P) * C + c) * P + i % P; };
    constexpr int KE = 20;
    const bool fast = per == 1 && n <= 256 * KE;
    const float* xp = x + ((long)g * C + c) * P;
    const float* gp = gy + ((long)g * C + c) * P;
    float xv[KE], gv[KE];
    double s1 = 0, s2 = 0;
    if (fast) {
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const long i = threadIdx.x + 256 * k;
            xv[k] = i < n ? xp[i] : 0.f;
            gv[k] = i < n ? gp[i] : 0.f;
            const double d = gv[k];
            s1 += d;
            s2 += d * ((double)xv[k] - mean) * inv;          // (out-of-range slots hold d = 0)
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 256) {
            const long o = at(i);
            const double d = gy[o];
            s1 += d;
            s2 += d * ((double)x[o] - mean) * inv;
        }
    }
    {
        const double w1 = wave_sum(s1), w2 = wave_sum(s2);
        if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = w1; sh[1][threadIdx.x >> 6] = w2; }
        __syncthreads();
    }
    const double m1 = (((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3]) / (double)n, m2 = (((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3]) / (double)n;
    const double k_ = (w ? (double)w[c] : 1.0) * inv;
    if (fast) {
        float* op = gx + ((long)g * C + c) * P;
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const long i = threadIdx.x + 256 * k;
            if (i < n) op[i] = (float)(k_ * ((double)gv[k] - m1 - ((double)xv[k] - mean) * inv * m2));
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 256) {
            const long o = at(i);
            gx[o] = (float)(k_ * ((double)gy[o] - m1 - ((double)x[o] - mean) * inv * m2));
        }
    }
}

// ---- svd_reg (dm/DFM_pose_refine.py:119-129): the 3x3 block A of a regressed pose replaced by U V^T of its SVD ------------------------
// = the orthogonal polar factor of A.  One thread per pose, float64 inside: one-sided Jacobi on the columns of A (B = A V, V a
// product of plane rotations, until the columns are orthogonal; sigma_j = |B_j|, U_j = B_j / sigma_j).  U V^T does not depend on the
// order or signs of the singular triplets, which is all the reference uses.  The backward is the derivative of the polar factor,
//     d A = U [ (H - H^T) o K ] V^T,   H = U^T G V,   K_ij = 1 / (sigma_i + sigma_j)
// -- well conditioned for the near-rotations a pose network regresses (sigma ~ 1: K ~ 1/2), where autograd through torch.svd divides
// by sigma_i^2 - sigma_j^2 ~ 0 and the two halves of U V^T cancel (the loop-gradient excess of mode 2 lived there).
struct Svd3 {
    double U[3][3], V[3][3], s[3];
};

__device__ void svd3(const double A[3][3], Svd3& o) {
    double B[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { B[i][j] = A[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += B[i][p] * B[i][p]; be += B[i][q] * B[i][q]; ga += B[i][p] * B[i][q]; }
                const double lim = 1e-17 * sqrt(al * be);
                if (fabs(ga) <= lim || ga == 0.0) continue;
                off = fmax(off, fabs(ga) / sqrt(al * be));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double bp = B[i][p], bq = B[i][q];
                    B[i][p] = c * bp - sn * bq;
                    B[i][q] = sn * bp + c * bq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq;
                    V[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double smax = 0;
    for (int j = 0; j < 3; ++j) {
        o.s[j] = sqrt(B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j]);
        smax = fmax(smax, o.s[j]);
    }
    // U: normalised columns; a column of (numerically) zero length is completed to a right-handed frame with the others
    bool ok[3];
    for (int j = 0; j < 3; ++j) {
        ok[j] = o.s[j] > 1e-300 && o.s[j] > 1e-14 * smax;
        for (int i = 0; i < 3; ++i) o.U[i][j] = ok[j] ? B[i][j] / o.s[j] : 0.0;
    }
    for (int j = 0; j < 3; ++j)
        if (!ok[j]) {
            const int a = (j + 1) % 3, b = (j + 2) % 3;
            if (!ok[a] || !ok[b]) {                     // rank <= 1: any completion; take coordinate axes not parallel to what exists
                for (int i = 0; i < 3; ++i) o.U[i][j] = i == j ? 1.0 : 0.0;
                continue;
            }
            o.U[0][j] = o.U[1][a] * o.U[2][b] - o.U[2][a] * o.U[1][b];
            o.U[1][j] = o.U[2][a] * o.U[0][b] - o.U[0][a] * o.U[2][b];
            o.U[2][j] = o.U[0][a] * o.U[1][b] - o.U[1][a] * o.U[0][b];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o.V[i][j] = V[i][j];
}

// pose [n,3,4] -> out [n,3,4] (rotation block U V^T, translation column copied); save [n,21] doubles = U, V (row-major), sigma
// WORLD (nefes_regressed_pose_*): the translation column leaves as t * t_scale + t_move (fix_coord_supp, dm/direct_pose_model.py:210-232,
// in torch's fp32 operation order: one multiply, one add); do_svd = 0 copies the rotation block (svd_reg off).
struct WorldT {
    float scale, mx, my, mz;
};
template <bool WORLD>
__global__ void svd_reg_fwd_kernel(int n, const float* __restrict__ pose, float* __restrict__ out, double* __restrict__ save, int do_svd, WorldT wt) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* p = pose + (long)k * 12;
    float* o = out + (long)k * 12;
    if (WORLD && !do_svd) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) o[i * 4 + j] = p[i * 4 + j];
            o[i * 4 + 3] = __fadd_rn(__fmul_rn(p[i * 4 + 3], wt.scale), i == 0 ? wt.mx : (i == 1 ? wt.my : wt.mz));
        }
        return;
    }
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = p[i * 4 + j];
    Svd3 d;
    svd3(A, d);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = (float)(d.U[i][0] * d.V[j][0] + d.U[i][1] * d.V[j][1] + d.U[i][2] * d.V[j][2]);
        if (WORLD) o[i * 4 + 3] = __fadd_rn(__fmul_rn(p[i * 4 + 3], wt.scale), i == 0 ? wt.mx : (i == 1 ? wt.my : wt.mz));
        else o[i * 4 + 3] = p[i * 4 + 3];
    }
    if (save) {
        double* sv = save + (long)k * 21;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { sv[i * 3 + j] = d.U[i][j]; sv[9 + i * 3 + j] = d.V[i][j]; }
        for (int j = 0; j < 3; ++j) sv[18 + j] = d.s[j];
    }
}

template <bool WORLD>
__global__ void svd_reg_bwd_kernel(int n, const double* __restrict__ save, const float* __restrict__ g_out, float* __restrict__ g_pose, int do_svd,
                                   float t_scale) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double* sv = save + (long)k * 21;
    const float* g = g_out + (long)k * 12;
    if (WORLD && !do_svd) {
        float* o_ = g_pose + (long)k * 12;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) o_[i * 4 + j] = g[i * 4 + j];
            o_[i * 4 + 3] = g[i * 4 + 3] * t_scale;
        }
        return;
    }
    double U[3][3], V[3][3], s[3], GV[3][3], H[3][3], M[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { U[i][j] = sv[i * 3 + j]; V[i][j] = sv[9 + i * 3 + j]; }
    for (int j = 0; j < 3; ++j) s[j] = sv[18 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) GV[i][j] = (double)g[i * 4 + 0] * V[0][j] + (double)g[i * 4 + 1] * V[1][j] + (double)g[i * 4 + 2] * V[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[i][j] = U[0][i] * GV[0][j] + U[1][i] * GV[1][j] + U[2][i] * GV[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double den = s[i] + s[j];
            M[i][j] = den > 0 ? (H[i][j] - H[j][i]) / den : 0.0;          // (rank <= 1: the polar factor is not differentiable there)
        }
    float* o = g_pose + (long)k * 12;
    for (int i = 0; i < 3; ++i) {
        double UM[3];
        for (int j = 0; j < 3; ++j) UM[j] = U[i][0] * M[0][j] + U[i][1] * M[1][j] + U[i][2] * M[2][j];
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = (float)(UM[0] * V[j][0] + UM[1] * V[j][1] + UM[2] * V[j][2]);
        o[i * 4 + 3] = WORLD ? g[i * 4 + 3] * t_scale : g[i * 4 + 3];
    }
}

// ---- cosine feature loss -------------------------------------------------------------------------------------------------
constexpr int kParts = 8;      // blocks per channel

__global__ __launch_bounds__(256) void cosine_partial_kernel(int C, long P, const float* __restrict__ a, const float* __restrict__ b,
                                                             double* __restrict__ part) {      // part [C][kParts][3]
    const int c = blockIdx.x / kParts, q = blockIdx.x % kParts;
    const long per = (P + kParts - 1) / kParts, lo = q * per, hi = lo + per < P ? lo + per : P;
    const float* pa = a + (long)c * P;
    const float* pb = b + (long)c * P;
    double dab = 0, daa = 0, dbb = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const double x = pa[i], y = pb[i];
        dab += x * y;
        daa += x * x;
        dbb += y * y;
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = dab; sh[1][threadIdx.x] = daa; sh[2][threadIdx.x] = dbb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + ((long)c * kParts + q) * 3;
        o[0] = sh[0][0]; o[1] = sh[1][0]; o[2] = sh[2][0];
    }
}

// stats [C][4] = (dot, |a|, |b|, cos); loss = 1 - mean cos
__global__ __launch_bounds__(256) void cosine_final_kernel(int C, double eps, const double* __restrict__ part, double* __restrict__ stats,
                                                           float* __restrict__ loss) {
    __shared__ double sh[256];
    double acc = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        double dab = 0, daa = 0, dbb = 0;
        for (int q = 0; q < kParts; ++q) {
            const double* o = part + ((long)c * kParts + q) * 3;
            dab += o[0]; daa += o[1]; dbb += o[2];
        }
        const double na = sqrt(daa), nb = sqrt(dbb);
        const double cs = dab / ((na > eps ? na : eps) * (nb > eps ? nb : eps));
        stats[c * 4 + 0] = dab; stats[c * 4 + 1] = na; stats[c * 4 + 2] = nb; stats[c * 4 + 3] = cs;
        acc += cs;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(1.0 - sh[0] / C);
}

// g_a[c][p] = -g_loss / C * d cos_c / d a[c][p]
__global__ __launch_bounds__(256) void cosine_bwd_kernel(int C, long P, double eps, const float* __restrict__ a, const float* __restrict__ b,
                                                         const double* __restrict__ stats, const float* __restrict__ g_loss,
                                                         float* __restrict__ g_a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)C * P) return;
    const int c = (int)(idx / P);
    const double dab = stats[c * 4 + 0], na = stats[c * 4 + 1], nb = stats[c * 4 + 2];
    const double nbc = nb > eps ? nb : eps;
    const double k = -(double)g_loss[0] / C;
    double v;
    if (na > eps) v = (double)b[idx] / (na * nbc) - dab * (double)a[idx] / (na * na * na * nbc);
    else v = (double)b[idx] / (eps * nbc);                                   // the clamped norm is a constant there
    g_a[idx] = (float)(k * v);
}

// ---- feature loss ON the up-sampled image, without the up-sampled image (DFM_APR_refine.py:114-131): the loop up-samples the
// fused [C,h,w] features bicubically to (H, W), crops 10 px and takes the cosine loss against the query image's features.  As
// separate kernels that is 34 MB written and read again each way (up-sampled features; their gradient) around 2.4 MB of data.
// Here every up-sampled value is interpolated where it is consumed -- same expression, same order as bicubic_up_fwd_kernel -- from
// the channel's low-resolution plane (forward: staged in LDS) or its four source rows (backward), and the target is the only
// large tensor that moves.
struct UpCosArgs {
    int C, h, w, OH, OW, o0, CH, CW;
    float sy, sx;
    const float* x;        // [C, h, w]
    const float* target;   // [C, CH, CW]
};

__device__ __forceinline__ float up_value(const float* plane, int w, int h, const nefes_bicubic::Taps& ty, const nefes_bicubic::Taps& tx) {
    using namespace nefes_bicubic;
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = clampi(tx.base - 1 + j, w);
    float rows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* r = plane + (long)clampi(ty.base - 1 + i, h) * w;
        rows[i] = r[xs[0]] * tx.w[0] + r[xs[1]] * tx.w[1] + r[xs[2]] * tx.w[2] + r[xs[3]] * tx.w[3];
    }
    return rows[0] * ty.w[0] + rows[1] * ty.w[1] + rows[2] * ty.w[2] + rows[3] * ty.w[3];
}

__global__ __launch_bounds__(256) void upcos_partial_kernel(UpCosArgs a, double* __restrict__ part) {      // part [C][kParts][3]
    using namespace nefes_bicubic;
    extern __shared__ float plane[];
    const int c = blockIdx.x / kParts, q = blockIdx.x % kParts;
    const float* src = a.x + (long)c * a.h * a.w;
    for (int i = threadIdx.x; i < a.h * a.w; i += 256) plane[i] = src[i];
    __syncthreads();
    // part q = a band of up-sampled rows; a thread owns columns (its x taps and clamped source columns are computed once)
    const int per = (a.CH + kParts - 1) / kParts, r_lo = q * per, r_hi = r_lo + per < a.CH ? r_lo + per : a.CH;
    double dab = 0, daa = 0, dbb = 0;
    for (int j = threadIdx.x; j < a.CW; j += 256) {
        const Taps tx = taps_of(a.sx, j + a.o0);
        int xs[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xs[m] = clampi(tx.base - 1 + m, a.w);
        const float* pb = a.target + (long)c * a.CH * a.CW + j;
        for (int r = r_lo; r < r_hi; ++r) {
            const Taps ty = taps_of(a.sy, r + a.o0);
            float rows[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* rr = plane + clampi(ty.base - 1 + i, a.h) * a.w;
                rows[i] = rr[xs[0]] * tx.w[0] + rr[xs[1]] * tx.w[1] + rr[xs[2]] * tx.w[2] + rr[xs[3]] * tx.w[3];
            }
            const double xv = rows[0] * ty.w[0] + rows[1] * ty.w[1] + rows[2] * ty.w[2] + rows[3] * ty.w[3];
            const double yv = pb[(long)r * a.CW];
            dab += xv * yv;
            daa += xv * xv;
            dbb += yv * yv;
        }
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = dab; sh[1][threadIdx.x] = daa; sh[2][threadIdx.x] = dbb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + ((long)c * kParts + q) * 3;
        o[0] = sh[0][0]; o[1] = sh[1][0]; o[2] = sh[2][0];
    }
}

// One workgroup per (channel, up-sampled row of the window): the row's gradient d loss / d up[c][oy][:] is formed in LDS from the
// interpolated values and the target row, and gathered along x onto the w source columns: tmp[c][oy - o0][0..w).  The gather
// along y (bicubic_gather_kernel of upsample.hip, inner = w) finishes the job.
__global__ __launch_bounds__(128) void upcos_bwd_rows_kernel(UpCosArgs a, double eps, const double* __restrict__ stats,
                                                             const float* __restrict__ g_loss, nefes_bicubic::GatherTable gx,
                                                             float* __restrict__ tmp) {
    using namespace nefes_bicubic;
    extern __shared__ float lds[];
    float* rows = lds;                 // [4][w] the four source rows of this up-sampled row
    float* gs = lds + 4 * a.w;         // [CW]
    const int c = blockIdx.x / a.CH, r = blockIdx.x % a.CH, oy = r + a.o0;
    const Taps ty = taps_of(a.sy, oy);
    const float* src = a.x + (long)c * a.h * a.w;
    for (int i = threadIdx.x; i < 4 * a.w; i += 128) rows[i] = src[(long)clampi(ty.base - 1 + i / a.w, a.h) * a.w + i % a.w];
    __syncthreads();
    const double dab = stats[c * 4 + 0], na = stats[c * 4 + 1], nb = stats[c * 4 + 2];
    const double nbc = nb > eps ? nb : eps;
    const double k = -(double)g_loss[0] / a.C;
    // d cos / d a = b / (|a||b|) - (a.b) a / (|a|^3 |b|)  (a clamped norm is a constant): two factors per channel, not per pixel
    const double k1 = na > eps ? k / (na * nbc) : k / (eps * nbc), k2 = na > eps ? k * dab / (na * na * na * nbc) : 0.0;
    const float* pb = a.target + ((long)c * a.CH + r) * a.CW;
    for (int j = threadIdx.x; j < a.CW; j += 128) {
        const Taps tx = taps_of(a.sx, j + a.o0);
        int xs[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xs[m] = clampi(tx.base - 1 + m, a.w);
        float rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* rr = rows + i * a.w;
            rv[i] = rr[xs[0]] * tx.w[0] + rr[xs[1]] * tx.w[1] + rr[xs[2]] * tx.w[2] + rr[xs[3]] * tx.w[3];
        }
        const double av = rv[0] * ty.w[0] + rv[1] * ty.w[1] + rv[2] * ty.w[2] + rv[3] * ty.w[3];
        gs[j] = (float)(k1 * (double)pb[j] - k2 * av);
    }
    __syncthreads();
    float* out = tmp + ((long)c * a.CH + r) * a.w;
    for (int xi = threadIdx.x; xi < a.w; xi += 128) out[xi] = gather_axis_table(gx, xi, gs, 1);
}

__global__ __launch_bounds__(256) void upcos_gather_rows_kernel(long outer, int n_in, int n_win, long inner, nefes_bicubic::GatherTable gy,
                                                                const float* __restrict__ src, float* __restrict__ dst) {
    using namespace nefes_bicubic;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= outer * n_in * inner) return;
    const long q = idx % inner;
    const int y = (int)((idx / inner) % n_in);
    const long p = idx / (inner * n_in);
    dst[idx] = gather_axis_table(gy, y, src + p * n_win * inner + q, inner);
}


// ---- the same loss for a FIXED target, through the Gram matrices of the up-sampling (round 5) ---------------------------------
// Per channel the up-sampled, cropped image is  up = Uy X Ux^T  (Uy [CH, h], Ux [CW, w]: the bicubic taps of the window's rows and
// columns, clamped at the borders).  The refinement loop holds the target for all iterations of an image (DFM_APR_refine.py:
// 100-131: the query image's features are extracted once, the pose is optimised opt_iter times against them), so
//     <up, target> = <X, Uy^T target Ux> = <X, Tt>            Tt [C, h, w]: once per image (upcos_prep_* below)
//     |up|^2       = <X, Gy X Gx>                             Gy = Uy^T Uy [h, h], Gx = Ux^T Ux [w, w]: once per geometry, banded
//     |target|^2                                              once per image
//     d loss / d X = -g / C (k1 Tt - k2 Gy X Gx)              (k1, k2 of upcos_bwd_rows_kernel)
// and an iteration touches the 2.4 MB of X, Tt and P = Gy X Gx instead of the 34 MB target twice: 30 + 47 + 14 us -> three small
// launches.  Everything in float64 from the float32 tap weights (those of taps_of: what ATen's kernel uses), so the result is the
// float64 value of the loss on the float32 taps -- the one-pass kernels above round every up-sampled value to float32 first.
__global__ void bicubic_gram_kernel(int n_in, int o0, int n_win, float scale, double* __restrict__ G) {
    using namespace nefes_bicubic;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_in * n_in) return;
    const int ya = idx / n_in, yb = idx % n_in;
    double acc = 0;
    for (int o = o0; o < o0 + n_win; ++o) {
        const Taps t = taps_of(scale, o);
        double wa = 0, wb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = clampi(t.base - 1 + k, n_in);
            if (y == ya) wa += (double)t.w[k];
            if (y == yb) wb += (double)t.w[k];
        }
        acc += wa * wb;
    }
    G[idx] = acc;
}

// tmp[c][r][x] = sum_ox Wx(ox -> x) target[c][r][ox]
__global__ __launch_bounds__(256) void upcos_prep_rows_kernel(long n, int CW, int w, nefes_bicubic::GatherTable gx,
                                                              const float* __restrict__ target, double* __restrict__ tmp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % w);
    const long row = idx / w;                                    // c * CH + r
    const int f = gx.first[x], cnt = gx.count[x];
    const float* wt = gx.wt + (long)x * gx.T;
    const float* s = target + row * CW + f;
    double acc = 0;
    for (int i = 0; i < cnt; ++i) acc += (double)wt[i] * (double)s[i];
    tmp[idx] = acc;
}

// tt[c][y][x] = sum_r Wy(r -> y) tmp[c][r][x]
__global__ __launch_bounds__(256) void upcos_prep_cols_kernel(long n, int h, int CH, int w, nefes_bicubic::GatherTable gy,
                                                              const double* __restrict__ tmp, double* __restrict__ tt) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h);
    const long c = idx / ((long)w * h);
    const int f = gy.first[y], cnt = gy.count[y];
    const float* wt = gy.wt + (long)y * gy.T;
    const double* s = tmp + (c * CH + f) * w + x;
    double acc = 0;
    for (int i = 0; i < cnt; ++i) acc += (double)wt[i] * s[(long)i * w];
    tt[idx] = acc;
}

// dbb[c] = |target[c]|^2
__global__ __launch_bounds__(256) void upcos_prep_norm_kernel(long P, const float* __restrict__ target, double* __restrict__ dbb) {
    const float* pb = target + (long)blockIdx.x * P;
    double acc = 0;
    for (long i = threadIdx.x; i < P; i += 256) {
        const double v = pb[i];
        acc += v * v;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbb[blockIdx.x] = sh[0];
}

// One workgroup per (channel, band of source rows): Q = X Gx on the band's rows and R more on either side (LDS), P = Gy Q on the
// band, partial <X, Tt> and <X, P> in the layout cosine_final_kernel sums (|target|^2 rides in part 0).  Fixed order: deterministic.
__global__ __launch_bounds__(256) void upcos_gram_fwd_kernel(int h, int w, int R, const float* __restrict__ x, const double* __restrict__ tt,
                                                             const double* __restrict__ dbb, const double* __restrict__ Gx,
                                                             const double* __restrict__ Gy, double* __restrict__ part,
                                                             double* __restrict__ pmat) {
    extern __shared__ double lds_d[];
    const int c = blockIdx.x / kParts, q = blockIdx.x % kParts;
    const int per = (h + kParts - 1) / kParts, r_lo = q * per < h ? q * per : h, r_hi = r_lo + per < h ? r_lo + per : h;
    const int y_lo = r_lo - R > 0 ? r_lo - R : 0, y_hi = r_hi + R < h ? r_hi + R : h, nrow = r_hi > r_lo ? y_hi - y_lo : 0;
    const int B = 2 * R + 1;
    double* Q = lds_d;                                  // [nrow][w]
    double* gxb = Q + (per + 2 * R) * w;                // [w][B]: column xx of Gx, rows xx - R .. xx + R (zero outside the matrix)
    double* gyb = gxb + w * B;                          // [per][B]: row y of Gy, columns y - R .. y + R
    float* X = (float*)(gyb + per * B);                 // [nrow][w]
    const float* src = x + ((long)c * h + y_lo) * w;
    // everything that comes from memory in ONE round: the rows of X, the band of Gx, this part's rows of Gy's band
    for (int i = threadIdx.x; i < nrow * w; i += 256) X[i] = src[i];
    for (int i = threadIdx.x; i < w * B; i += 256) {
        const int xx = i / B, a = xx - R + i % B;
        gxb[i] = a >= 0 && a < w ? Gx[(long)a * w + xx] : 0.0;
    }
    for (int i = threadIdx.x; i < (r_hi - r_lo) * B; i += 256) {
        const int y = r_lo + i / B, b = y - R + i % B;
        gyb[i] = b >= 0 && b < h ? Gy[(long)y * h + b] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nrow * w; i += 256) {
        const int xx = i % w, yy = i / w;
        const int a0 = xx - R > 0 ? xx - R : 0, a1 = xx + R < w - 1 ? xx + R : w - 1;
        double acc = 0;
        for (int a = a0; a <= a1; ++a) acc += (double)X[yy * w + a] * gxb[xx * B + a - xx + R];
        Q[i] = acc;
    }
    __syncthreads();
    double dab = 0, daa = 0;
    for (int i = threadIdx.x; i < (r_hi - r_lo) * w; i += 256) {
        const int xx = i % w, y = r_lo + i / w;
        const int b0 = y - R > 0 ? y - R : 0, b1 = y + R < h - 1 ? y + R : h - 1;
        const long o = ((long)c * h + y) * w + xx;
        const double tv = tt[o];
        double p = 0;
        for (int b = b0; b <= b1; ++b) p += gyb[(y - r_lo) * B + b - y + R] * Q[(b - y_lo) * w + xx];
        const double xv = X[(y - y_lo) * w + xx];
        dab += xv * tv;
        daa += xv * p;
        pmat[o] = p;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = dab; sh[1][threadIdx.x] = daa;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + ((long)c * kParts + q) * 3;
        o[0] = sh[0][0]; o[1] = sh[1][0]; o[2] = q == 0 ? dbb[c] : 0.0;
    }
}

// g_x = -g / C (k1 Tt - k2 P): the factors of upcos_bwd_rows_kernel
__global__ __launch_bounds__(256) void upcos_gram_bwd_kernel(int C, long P, double eps, const double* __restrict__ tt, const double* __restrict__ pmat,
                                                             const double* __restrict__ stats, const float* __restrict__ g_loss,
                                                             float* __restrict__ g_x) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)C * P) return;
    const int c = (int)(idx / P);
    const double dab = stats[c * 4 + 0], na = stats[c * 4 + 1], nb = stats[c * 4 + 2];
    const double nbc = nb > eps ? nb : eps;
    const double k = -(double)g_loss[0] / C;
    const double k1 = na > eps ? k / (na * nbc) : k / (eps * nbc), k2 = na > eps ? k * dab / (na * na * na * nbc) : 0.0;
    g_x[idx] = (float)(k1 * tt[idx] - k2 * pmat[idx]);
}

// ---- verification step of train_on_batch (DFM_APR_refine.py:117-128, :146-150): PSNR and mean SSIM of two [C,H,W] images --------------
// SSIM = utils/utils.py:15-49: 7x7 mean filters (nn.AvgPool2d(7, 1)) over the ReflectionPad2d(3) images, clamp(n / d, 0, 1).  In torch
// that is two pads, five pools and ~15 element-wise launches per call; here one thread per pixel sums its 49-pixel windows of x, y,
// x^2, y^2, xy (fp32, row-major, then / 49 as avg_pool2d does), float64 block partials, one fixed-order final.  x, y: [C] planes of
// H x W with a row and a plane stride in elements, so the loop's 10-pixel crop is a view, not a copy.
constexpr int kSsimBlock = 256;
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(kSsimBlock) void psnr_ssim_partial_kernel(int C, int H, int W, const float* __restrict__ x, long xr, long xc,
                                                                       const float* __restrict__ y, long yr, long yc,
                                                                       double* __restrict__ part) {        // part [blocks][2]
    const long P = (long)H * W, n = (long)C * P;
    double s_ssim = 0, s_se = 0;
    for (long i = (long)blockIdx.x * kSsimBlock + threadIdx.x; i < n; i += (long)gridDim.x * kSsimBlock) {
        const int c = (int)(i / P), r = (int)((i % P) / W), q = (int)(i % W);
        const float* px = x + c * xc;
        const float* py = y + c * yc;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dr = -3; dr <= 3; ++dr) {
            const int rr = reflect_idx(r + dr, H);
#pragma unroll
            for (int dq = -3; dq <= 3; ++dq) {
                const int qq = reflect_idx(q + dq, W);
                const float a = px[rr * xr + qq], b = py[rr * yr + qq];
                sx += a; sy += b;
                sxx = __fadd_rn(sxx, __fmul_rn(a, a)); syy = __fadd_rn(syy, __fmul_rn(b, b)); sxy = __fadd_rn(sxy, __fmul_rn(a, b));
            }
        }
        const float k = 49.f;
        const float mx = sx / k, my = sy / k;
        const float vx = __fsub_rn(sxx / k, __fmul_rn(mx, mx)), vy = __fsub_rn(syy / k, __fmul_rn(my, my)),
                    vxy = __fsub_rn(sxy / k, __fmul_rn(mx, my));
        const float c1 = (float)(0.01 * 0.01), c2 = (float)(0.03 * 0.03);      // (Python's 0.01 ** 2 as the fp32 scalar torch makes of it)
        const float nn = __fmul_rn(__fadd_rn(__fmul_rn(__fmul_rn(2.f, mx), my), c1), __fadd_rn(__fmul_rn(2.f, vxy), c2));
        const float dd = __fmul_rn(__fadd_rn(__fadd_rn(__fmul_rn(mx, mx), __fmul_rn(my, my)), c1), __fadd_rn(__fadd_rn(vx, vy), c2));
        float v = nn / dd;
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);          // (NaN stays NaN, as torch.clamp leaves it)
        s_ssim += (double)v;
        const float e = __fsub_rn(px[r * xr + q], py[r * yr + q]);
        s_se += (double)__fmul_rn(e, e);
    }
    __shared__ double sh[2][kSsimBlock];
    sh[0][threadIdx.x] = s_ssim; sh[1][threadIdx.x] = s_se;
    __syncthreads();
    for (int s2 = kSsimBlock / 2; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s2]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = sh[0][0]; part[blockIdx.x * 2 + 1] = sh[1][0]; }
}
// out[0] = psnr = -10 log10(mean squared error) (models/nerfh.py mse2psnr(img2mse(.))), out[1] = mean SSIM
__global__ void psnr_ssim_final_kernel(int blocks, double n, const double* __restrict__ part, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double a = 0, b = 0;
    for (int i = 0; i < blocks; ++i) { a += part[i * 2]; b += part[i * 2 + 1]; }
    out[0] = (float)(-10.0 * log(b / n) / log(10.0));
    out[1] = (float)(a / n);
}
int psnr_ssim_blocks(int C, int H, int W) {
    const long n = (long)C * H * W, b = (n + kSsimBlock - 1) / kSsimBlock;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t nefes_psnr_ssim_workspace(int C, int H, int W) {
    return C > 0 && H > 0 && W > 0 ? (size_t)psnr_ssim_blocks(C, H, W) * 2 * sizeof(double) : 0;
}
extern "C" int nefes_psnr_ssim(int C, int H, int W, const float* x, int64_t x_row_stride, int64_t x_plane_stride, const float* y,
                               int64_t y_row_stride, int64_t y_plane_stride, void* workspace, float* out, void* stream) {
    if (C <= 0 || H < 4 || W < 4 || !x || !y || !workspace || !out || x_row_stride < W || y_row_stride < W) return NEFES_E_BADARG;   // (reflection by 3 needs 4 pixels)
    const int nb = psnr_ssim_blocks(C, H, W);
    hipLaunchKernelGGL(psnr_ssim_partial_kernel, dim3(nb), dim3(kSsimBlock), 0, (hipStream_t)stream, C, H, W, x, (long)x_row_stride,
                       (long)x_plane_stride, y, (long)y_row_stride, (long)y_plane_stride, (double*)workspace);
    hipLaunchKernelGGL(psnr_ssim_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nb, (double)C * H * W, (const double*)workspace, out);
    return (int)hipGetLastError();
}

extern "C" int nefes_upcos_loss_fwd(int C, int h, int w, int OH, int OW, int crop, const float* x, const float* target, double* scratch,
                                    float* loss, void* stream) {
    const int CH = OH - 2 * crop, CW = OW - 2 * crop;
    if (C <= 0 || h <= 0 || w <= 0 || crop < 0 || CH <= 0 || CW <= 0 || !x || !target || !scratch || !loss) return NEFES_E_BADARG;
    if ((size_t)h * w * 4 > 96 * 1024) return NEFES_E_UNSUPPORTED;           // the channel's plane is staged in LDS
    UpCosArgs a{C, h, w, OH, OW, crop, CH, CW, (float)h / OH, (float)w / OW, x, target};
    double* part = scratch + (size_t)C * 4;
    const size_t lds = (size_t)h * w * 4;
    hipError_t e = hipFuncSetAttribute((const void*)upcos_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(upcos_partial_kernel, dim3(C * kParts), dim3(256), lds, (hipStream_t)stream, a, part);
    hipLaunchKernelGGL(cosine_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, C, 1e-6, (const double*)part, scratch, loss);
    return (int)hipGetLastError();
}

extern "C" int nefes_upcos_loss_bwd(int C, int h, int w, int OH, int OW, int crop, const float* x, const float* target, const double* scratch,
                                    const float* g_loss, const int* tx_first, const int* tx_count, const float* tx_wt, const int* ty_first,
                                    const int* ty_count, const float* ty_wt, int T, float* tmp, float* g_x, void* stream) {
    const int CH = OH - 2 * crop, CW = OW - 2 * crop;
    if (C <= 0 || h <= 0 || w <= 0 || crop < 0 || CH <= 0 || CW <= 0 || !x || !target || !scratch || !g_loss || !tmp || !g_x) return NEFES_E_BADARG;
    if (!tx_first || !tx_count || !tx_wt || !ty_first || !ty_count || !ty_wt || T <= 0) return NEFES_E_BADARG;
    const nefes_bicubic::GatherTable gx{tx_first, tx_count, tx_wt, T}, gy{ty_first, ty_count, ty_wt, T};
    UpCosArgs a{C, h, w, OH, OW, crop, CH, CW, (float)h / OH, (float)w / OW, x, target};
    const size_t lds = (size_t)(4 * w + CW) * 4;
    if (lds > 96 * 1024) return NEFES_E_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute((const void*)upcos_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(upcos_bwd_rows_kernel, dim3((unsigned)((long)C * CH)), dim3(128), lds, (hipStream_t)stream, a, 1e-6, scratch, g_loss, gx, tmp);
    const long n = (long)C * h * w;                    // g_x[c][y][x] = sum over the window's rows oy of Wy(oy -> y) tmp[c][oy - o0][x]
    hipLaunchKernelGGL(upcos_gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)C, h, CH, (long)w, gy,
                       (const float*)tmp, g_x);
    return (int)hipGetLastError();
}

extern "C" int nefes_bicubic_gram(int n_in, int n_out, int o0, int n_win, double* G, void* stream) {
    if (n_in <= 0 || n_out <= 0 || o0 < 0 || n_win <= 0 || o0 + n_win > n_out || !G) return NEFES_E_BADARG;
    if ((long)n_in * n_in > (1 << 24)) return NEFES_E_UNSUPPORTED;
    hipLaunchKernelGGL(bicubic_gram_kernel, dim3((unsigned)((n_in * n_in + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_in, o0, n_win,
                       (float)n_in / n_out, G);
    return (int)hipGetLastError();
}

extern "C" int nefes_upcos_prepare(int C, int h, int w, int OH, int OW, int crop, const float* target, const int* tx_first, const int* tx_count,
                                   const float* tx_wt, const int* ty_first, const int* ty_count, const float* ty_wt, int T, double* tmp,
                                   double* tt, double* dbb, void* stream) {
    const int CH = OH - 2 * crop, CW = OW - 2 * crop;
    if (C <= 0 || h <= 0 || w <= 0 || crop < 0 || CH <= 0 || CW <= 0 || !target || !tmp || !tt || !dbb) return NEFES_E_BADARG;
    if (!tx_first || !tx_count || !tx_wt || !ty_first || !ty_count || !ty_wt || T <= 0) return NEFES_E_BADARG;
    const nefes_bicubic::GatherTable gx{tx_first, tx_count, tx_wt, T}, gy{ty_first, ty_count, ty_wt, T};
    const long n1 = (long)C * CH * w, n2 = (long)C * h * w;
    if ((n1 + 255) / 256 > 0x7fffffffl) return NEFES_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(upcos_prep_rows_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, n1, CW, w, gx, target, tmp);
    hipLaunchKernelGGL(upcos_prep_cols_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, n2, h, CH, w, gy, (const double*)tmp, tt);
    hipLaunchKernelGGL(upcos_prep_norm_kernel, dim3((unsigned)C), dim3(256), 0, st, (long)CH * CW, target, dbb);
    return (int)hipGetLastError();
}

// LDS bytes of one upcos_gram_fwd_kernel workgroup (a band of rows of one channel: Q = X Gx and the two Gram bands); the launch needs
// it within NEFES_UPCOS_GRAM_LDS_MAX.  Callers ask BEFORE they commit to the prepared-target form (ops.UpcosTarget.fits): the one-pass
// kernels (nefes_upcos_loss_fwd / _bwd) take any geometry.
#define NEFES_UPCOS_GRAM_LDS_MAX (96 * 1024)
extern "C" size_t nefes_upcos_gram_lds_bytes(int h, int w, int band) {
    if (h <= 0 || w <= 0 || band < 0) return 0;
    const int per = (h + kParts - 1) / kParts;
    return (size_t)(per + 2 * band) * w * 12 + (size_t)(w + per) * (2 * band + 1) * 8;
}

extern "C" int nefes_upcos_gram_fwd(int C, int h, int w, const float* x, const double* tt, const double* dbb, const double* gram_x,
                                    const double* gram_y, int band, double* scratch, double* pmat, float* loss, void* stream) {
    if (C <= 0 || h <= 0 || w <= 0 || band < 0 || !x || !tt || !dbb || !gram_x || !gram_y || !scratch || !pmat || !loss) return NEFES_E_BADARG;
    const size_t lds = nefes_upcos_gram_lds_bytes(h, w, band);
    if (lds > NEFES_UPCOS_GRAM_LDS_MAX) return NEFES_E_UNSUPPORTED;
    // the kernel's dynamic-LDS ceiling, once per device of this process (a constant: not state a caller could observe)
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)upcos_gram_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NEFES_UPCOS_GRAM_LDS_MAX);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    double* part = scratch + (size_t)C * 4;
    hipLaunchKernelGGL(upcos_gram_fwd_kernel, dim3((unsigned)(C * kParts)), dim3(256), lds, (hipStream_t)stream, h, w, band, x, tt, dbb, gram_x,
                       gram_y, part, pmat);
    hipLaunchKernelGGL(cosine_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, C, 1e-6, (const double*)part, scratch, loss);
    return (int)hipGetLastError();
}

extern "C" int nefes_upcos_gram_bwd(int C, int h, int w, const double* tt, const double* pmat, const double* scratch, const float* g_loss,
                                    float* g_x, void* stream) {
    if (C <= 0 || h <= 0 || w <= 0 || !tt || !pmat || !scratch || !g_loss || !g_x) return NEFES_E_BADARG;
    const long n = (long)C * h * w;
    hipLaunchKernelGGL(upcos_gram_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, (long)h * w, 1e-6, tt, pmat,
                       scratch, g_loss, g_x);
    return (int)hipGetLastError();
}

extern "C" int nefes_bn_train_fwd(int B, int C, int64_t P, int per_image, const float* x, const float* weight, const float* bias, double eps,
                                  double momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, double* save,
                                  void* stream) {
    if (B <= 0 || C <= 0 || P <= 0 || !x || !y || !save) return NEFES_E_BADARG;
    const int groups = per_image ? B : 1;
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3((unsigned)(groups * C)), dim3(256), 0, (hipStream_t)stream, B, C, (long)P, groups, x, weight, bias, eps,
                       momentum, running_mean, running_var, (long long*)num_batches_tracked, y, save);
    return (int)hipGetLastError();
}

extern "C" int nefes_bn_train_bwd(int B, int C, int64_t P, int per_image, const float* x, const float* weight, const double* save, const float* g_y,
                                  float* g_x, void* stream) {
    if (B <= 0 || C <= 0 || P <= 0 || !x || !save || !g_y || !g_x) return NEFES_E_BADARG;
    const int groups = per_image ? B : 1;
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3((unsigned)(groups * C)), dim3(256), 0, (hipStream_t)stream, B, C, (long)P, groups, x, weight, save, g_y, g_x);
    return (int)hipGetLastError();
}

extern "C" int nefes_svd_reg_fwd(int n_poses, const float* pose, float* out, double* save, void* stream) {
    if (n_poses <= 0 || !pose || !out) return NEFES_E_BADARG;
    hipLaunchKernelGGL(svd_reg_fwd_kernel<false>, dim3((unsigned)((n_poses + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_poses, pose, out, save, 1, WorldT{1.f, 0.f, 0.f, 0.f});
    return (int)hipGetLastError();
}

extern "C" int nefes_svd_reg_bwd(int n_poses, const double* save, const float* g_out, float* g_pose, void* stream) {
    if (n_poses <= 0 || !save || !g_out || !g_pose) return NEFES_E_BADARG;
    hipLaunchKernelGGL(svd_reg_bwd_kernel<false>, dim3((unsigned)((n_poses + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_poses, save, g_out, g_pose, 1, 1.f);
    return (int)hipGetLastError();
}

// train_on_batch's pose chain behind the regression network as ONE launch each way (DFM_APR_refine.py:91-97): svd_reg (optional) and
// fix_coord_supp's translation t' = t * t_scale + t_move.  In torch that chain is ~8 element-wise / cat launches forward and ~12 in
// autograd's backward: two thirds of what an iteration of the shipped default mode cost over the LearnPose mode.
extern "C" int nefes_regressed_pose_fwd(int n_poses, const float* pose, int do_svd, float t_scale, float mx, float my, float mz, float* out,
                                        double* save, void* stream) {
    if (n_poses <= 0 || !pose || !out || (do_svd && !save)) return NEFES_E_BADARG;
    hipLaunchKernelGGL(svd_reg_fwd_kernel<true>, dim3((unsigned)((n_poses + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_poses, pose, out, save,
                       do_svd, WorldT{t_scale, mx, my, mz});
    return (int)hipGetLastError();
}

extern "C" int nefes_regressed_pose_bwd(int n_poses, const double* save, int do_svd, float t_scale, const float* g_out, float* g_pose, void* stream) {
    if (n_poses <= 0 || !g_out || !g_pose || (do_svd && !save)) return NEFES_E_BADARG;
    hipLaunchKernelGGL(svd_reg_bwd_kernel<true>, dim3((unsigned)((n_poses + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_poses, save, g_out, g_pose,
                       do_svd, t_scale);
    return (int)hipGetLastError();
}

extern "C" int nefes_pose_compose_fwd(int n_poses, const float* r, const float* t, const float* init_c2w, float pose_scale,
                                      const float* move, float pose_scale2, float* c2w, void* stream) {
    if (n_poses <= 0 || !r || !t || !init_c2w || !move || !c2w) return NEFES_E_BADARG;
    PoseArgs p{r, t, init_c2w, pose_scale, pose_scale2, {move[0], move[1], move[2]}};
    hipLaunchKernelGGL(pose_compose_fwd_kernel, dim3(n_poses), dim3(64), 0, (hipStream_t)stream, p, c2w);
    return (int)hipGetLastError();
}

extern "C" int nefes_pose_compose_bwd(int n_poses, const float* r, const float* t, const float* init_c2w, float pose_scale,
                                      const float* move, float pose_scale2, const float* g_c2w, float* g_r, float* g_t, void* stream) {
    if (n_poses <= 0 || !r || !t || !init_c2w || !move || !g_c2w || !g_r || !g_t) return NEFES_E_BADARG;
    PoseArgs p{r, t, init_c2w, pose_scale, pose_scale2, {move[0], move[1], move[2]}};
    hipLaunchKernelGGL(pose_compose_bwd_kernel, dim3(n_poses), dim3(64), 0, (hipStream_t)stream, p, g_c2w, g_r, g_t);
    return (int)hipGetLastError();
}

extern "C" size_t nefes_cosine_loss_scratch_doubles(int C) { return C > 0 ? (size_t)C * (kParts * 3 + 4) : 0; }

extern "C" int nefes_cosine_loss_fwd(int C, int64_t P, const float* a, const float* b, double* scratch, float* loss, void* stream) {
    if (C <= 0 || P <= 0 || !a || !b || !scratch || !loss) return NEFES_E_BADARG;
    double* part = scratch + (size_t)C * 4;
    hipLaunchKernelGGL(cosine_partial_kernel, dim3(C * kParts), dim3(256), 0, (hipStream_t)stream, C, (long)P, a, b, part);
    hipLaunchKernelGGL(cosine_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, C, 1e-6, (const double*)part, scratch, loss);
    return (int)hipGetLastError();
}

extern "C" int nefes_cosine_loss_bwd(int C, int64_t P, const float* a, const float* b, const double* scratch, const float* g_loss,
                                     float* g_a, void* stream) {
    if (C <= 0 || P <= 0 || !a || !b || !scratch || !g_loss || !g_a) return NEFES_E_BADARG;
    const long n = (long)C * P;
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, (long)P, 1e-6, a, b,
                       scratch, g_loss, g_a);
    return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------------
//   fusion_input      what the loop does between compositing and FusionNet's first convolution (script/models/nerfh_nff.py:
//                     605-626 affine_color_transform with the image's 12 exposure coefficients already known, :578-603
//                     run_fusion_net's reshape / permute / cat, FusionNet.forward's colour normalisation :395-402):
//                         y = sigmoid(K rgb + b);  x[b, 0:3] = (y - mean) / std;  x[b, 3:] = feat^T           ([B, 3+C, H*W])
//                     In torch: a GEMM, five element-wise passes and a transposing cat forward, five passes backward -- eleven
//                     launches of 4-8 us around 4800 pixels.  Here one launch each way; y is kept for the backward.
//   adam              torch.optim.Adam's update (defaults, no weight decay / amsgrad) for a handful of numbers with a learning
//                     rate per element: the refinement loop's (r, t) with (lr_r, lr_t) are two parameter groups = four launches.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct FusionInArgs {
    int HW, C, has_affine;
    const float* rgb;      // [B*HW, 3]
    const float* feat;     // [B*HW, C]
    const float* affine;   // [B, 12] (3x3 kernel row-major, 3 bias) or null
    float mean[3], stdv[3];
    float* x;              // [B, 3+C, HW]
    float* y;              // [B*HW, 3] colours after the transform (saved for the backward)
    const float* g_x;      // backward: [B, 3+C, HW]
    float* g_rgb;          // [B*HW, 3]
    float* g_feat;         // [B*HW, C]
};

__global__ __launch_bounds__(256) void fusion_input_fwd_kernel(FusionInArgs a) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y, n0 = blockIdx.x * 64, t = threadIdx.x;
    const int HW = a.HW, C = a.C;
    const size_t pix0 = (size_t)b * HW + n0;
    float* xb = a.x + (size_t)b * (3 + C) * HW;
    for (int c0 = 0; c0 < C; c0 += 64) {
        // sixteen loads in flight, then the selects: with `ok ? load : 0` per element every load was followed by its own wait (sixteen
        // memory round trips in a row per chunk, seen in the disassembly; 13.6 us for 2.5 MB).  Rows / channels past the end read
        // element 0 of the image (valid memory) and are zeroed afterwards.
        float vals[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r * 4 + (t >> 6), col = t & 63;
            const bool ok = n0 + row < HW && c0 + col < C;
            vals[r] = a.feat[ok ? (pix0 + row) * C + c0 + col : (size_t)b * HW * C];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r * 4 + (t >> 6), col = t & 63;
            tile[row][col] = (n0 + row < HW && c0 + col < C) ? vals[r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = r * 4 + (t >> 6), px = t & 63;
            if (n0 + px < HW && c0 + ch < C) xb[(size_t)(3 + c0 + ch) * HW + n0 + px] = tile[px][ch];
        }
        __syncthreads();
    }
    if (t < 64 && n0 + t < HW) {
        const float* p = a.rgb + (pix0 + t) * 3;
        float v[3] = {p[0], p[1], p[2]};
        if (a.has_affine) {
            const float* k = a.affine + b * 12;
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float lin = ((k[3 * c] * v[0] + k[3 * c + 1] * v[1]) + k[3 * c + 2] * v[2]) + k[9 + c];
                o[c] = 1.f / (1.f + expf(-lin));
            }
            v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.y) a.y[(pix0 + t) * 3 + c] = v[c];
            xb[(size_t)c * HW + n0 + t] = (v[c] - a.mean[c]) / a.stdv[c];
        }
    }
}

__global__ __launch_bounds__(256) void fusion_input_bwd_kernel(FusionInArgs a) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y, n0 = blockIdx.x * 64, t = threadIdx.x;
    const int HW = a.HW, C = a.C;
    const size_t pix0 = (size_t)b * HW + n0;
    const float* gb = a.g_x + (size_t)b * (3 + C) * HW;
    if (a.g_feat) {
        for (int c0 = 0; c0 < C; c0 += 64) {
            float vals[16];                                  // loads first, selects afterwards (see fusion_input_fwd_kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = r * 4 + (t >> 6), px = t & 63;
                const bool ok = n0 + px < HW && c0 + ch < C;
                vals[r] = gb[ok ? (size_t)(3 + c0 + ch) * HW + n0 + px : (size_t)0];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = r * 4 + (t >> 6), px = t & 63;
                tile[px][ch] = (n0 + px < HW && c0 + ch < C) ? vals[r] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r * 4 + (t >> 6), col = t & 63;
                if (n0 + row < HW && c0 + col < C) a.g_feat[(pix0 + row) * C + c0 + col] = tile[row][col];
            }
            __syncthreads();
        }
    }
    if (a.g_rgb && t < 64 && n0 + t < HW) {
        float gy[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gy[c] = gb[(size_t)c * HW + n0 + t] / a.stdv[c];
        float o[3] = {gy[0], gy[1], gy[2]};
        if (a.has_affine) {
            const float* k = a.affine + b * 12;
            const float* y = a.y + (pix0 + t) * 3;
            float gl[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gl[c] = gy[c] * ((1.f - y[c]) * y[c]);          // sigmoid_backward: g * (1 - y) * y
#pragma unroll
            for (int j = 0; j < 3; ++j) o[j] = (k[j] * gl[0] + k[3 + j] * gl[1]) + k[6 + j] * gl[2];
        }
        float* g = a.g_rgb + (pix0 + t) * 3;
        g[0] = o[0]; g[1] = o[1]; g[2] = o[2];
    }
}

// torch.optim.Adam, single-tensor formulas of torch/optim/adam.py as the CPU reference evaluates them: the bias corrections and
// the step size are Python floats (float64) rounded to the tensor's dtype when they enter an operation:
//   m.lerp_(g, 1 - b1);  v.mul_(b2).addcmul_(g, g, value=1 - b2);  denom = (v.sqrt() / sqrt(1 - b2^t)).add_(eps);
//   p.addcdiv_(m, denom, value=-(lr / (1 - b1^t)))
__global__ void adam_kernel(int n, float* p, const float* g, float* m, float* v, float* step, const double* lr, double b1, double b2,
                            double eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float t = step[0] + 1.f;
    if (i < n) {
        const float gi = g[i];
        const float mi = m[i] + (float)(1.0 - b1) * (gi - m[i]);                 // torch.lerp, weight < 0.5
        const float vi = v[i] * (float)b2 + ((float)(1.0 - b2) * gi) * gi;
        m[i] = mi; v[i] = vi;
        const double c1 = 1.0 - pow(b1, (double)t), c2s = sqrt(1.0 - pow(b2, (double)t));
        const float denom = sqrtf(vi) / (float)c2s + (float)eps;
        p[i] = p[i] + (float)(-(lr[i] / c1)) * (mi / denom);
    }
    __syncthreads();
    if (i == 0) step[0] = t;
}

// The factored feature head's per-ray part (field_fwd_h3.hip FH; script/models/nerfh_nff.py:119-125 with the head of :487-490 pulled out of
// the sum): feat[n][c] = sum_f gmap[n][f] W[c][f] + gmap[n][F] b[c], gmap = the composited g (F values) and the composited ones channel.
// One workgroup per ray, the F + 1 inputs in LDS, every output a fixed-order sum over f: a ray's result does not depend on which other rays
// are in the batch (row shards, batched poses and single renders stay bit-identical -- a library GEMM picks its tiling by the batch size).
__global__ __launch_bounds__(128) void feat_head_fwd_kernel(int N, int C, int F, const float* __restrict__ gmap, const float* __restrict__ w_t,
                                                            const float* __restrict__ b, float* __restrict__ feat) {
    __shared__ float g[256];
    const int n = blockIdx.x;
    for (int f = threadIdx.x; f <= F; f += 128) g[f] = gmap[(size_t)n * (F + 1) + f];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 128) {
        // sixteen interleaved partial sums (f mod 16), combined in a fixed tree: the loads of sixteen terms are in flight together -- a single
        // chain of F dependent L2 round trips per output is what this launch's time was (17 us for 40 MFLOP)
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.f;
        int f = 0;
        for (; f + 15 < F; f += 16) {
            float wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = w_t[(size_t)(f + i) * C + c];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(g[f + i], wv[i], a[i]);
        }
        for (; f < F; ++f) a[0] = fmaf(g[f], w_t[(size_t)f * C + c], a[0]);
#pragma unroll
        for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
            for (int i = 0; i < st; ++i) a[i] += a[i + st];
        const float a0 = a[0], a1 = 0.f, a2 = 0.f, a3 = 0.f;
        feat[(size_t)n * C + c] = fmaf(g[F], b[c], (a0 + a1) + (a2 + a3));
    }
}
// its backward to gmap (W and b frozen): g_gmap[n][f] = sum_c g_feat[n][c] W[c][f]; g_gmap[n][F] = sum_c g_feat[n][c] b[c]
__global__ __launch_bounds__(128) void feat_head_bwd_kernel(int N, int C, int F, const float* __restrict__ g_feat, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ g_gmap) {
    __shared__ float g[256];
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 128) g[c] = g_feat[(size_t)n * C + c];
    __syncthreads();
    for (int f = threadIdx.x; f <= F; f += 128) {
        const float* col = f < F ? w + f : b;                       // column f of W (stride F), or the bias vector (stride 1)
        const size_t st_ = f < F ? (size_t)F : 1;
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.f;
        int c = 0;
        for (; c + 15 < C; c += 16) {
            float wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = col[(size_t)(c + i) * st_];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(g[c + i], wv[i], a[i]);
        }
        for (; c < C; ++c) a[0] = fmaf(g[c], col[(size_t)c * st_], a[0]);
#pragma unroll
        for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
            for (int i = 0; i < st; ++i) a[i] += a[i + st];
        const float a0 = a[0], a1 = 0.f, a2 = 0.f, a3 = 0.f;
        g_gmap[(size_t)n * (F + 1) + f] = (a0 + a1) + (a2 + a3);
    }
}

}  // namespace

extern "C" int nefes_feat_head_fwd(int N, int C, int F, const float* gmap, const float* w_t, const float* b, float* feat, void* stream) {
    if (N <= 0 || C <= 0 || C > 256 || F <= 0 || F >= 256 || !gmap || !w_t || !b || !feat) return NEFES_E_BADARG;
    hipLaunchKernelGGL(feat_head_fwd_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, N, C, F, gmap, w_t, b, feat);
    return (int)hipGetLastError();
}

extern "C" int nefes_feat_head_bwd(int N, int C, int F, const float* g_feat, const float* w, const float* b, float* g_gmap, void* stream) {
    if (N <= 0 || C <= 0 || C > 256 || F <= 0 || F >= 256 || !g_feat || !w || !b || !g_gmap) return NEFES_E_BADARG;
    hipLaunchKernelGGL(feat_head_bwd_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, N, C, F, g_feat, w, b, g_gmap);
    return (int)hipGetLastError();
}

extern "C" int nefes_fusion_input_fwd(int B, int HW, int C, const float* rgb, const float* feat, const float* affine, const float* mean3,
                                      const float* std3, float* x, float* y, void* stream) {
    if (B <= 0 || HW <= 0 || C < 0 || !rgb || !x || (C > 0 && !feat) || !mean3 || !std3 || (affine && !y)) return NEFES_E_BADARG;
    FusionInArgs a = {};
    a.HW = HW; a.C = C; a.has_affine = affine ? 1 : 0; a.rgb = rgb; a.feat = feat; a.affine = affine; a.x = x; a.y = y;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
    hipLaunchKernelGGL(fusion_input_fwd_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int nefes_fusion_input_bwd(int B, int HW, int C, const float* g_x, const float* affine, const float* y, const float* std3,
                                      float* g_rgb, float* g_feat, void* stream) {
    if (B <= 0 || HW <= 0 || C < 0 || !g_x || !std3 || (affine && !y) || (!g_rgb && !g_feat)) return NEFES_E_BADARG;
    FusionInArgs a = {};
    a.HW = HW; a.C = C; a.has_affine = affine ? 1 : 0; a.affine = affine; a.y = const_cast<float*>(y); a.g_x = g_x; a.g_rgb = g_rgb;
    a.g_feat = C > 0 ? g_feat : nullptr;
    for (int c = 0; c < 3; ++c) a.stdv[c] = std3[c];
    hipLaunchKernelGGL(fusion_input_bwd_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int nefes_adam_step(int n, float* p, const float* g, float* m, float* v, float* step, const double* lr, double beta1,
                               double beta2, double eps, void* stream) {
    if (n <= 0 || n > 1024 || !p || !g || !m || !v || !step || !lr) return NEFES_E_BADARG;
    hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, p, g, m, v, step, lr, beta1, beta2, eps);
    return (int)hipGetLastError();
}
