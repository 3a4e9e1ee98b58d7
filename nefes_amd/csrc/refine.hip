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

#include "../../include/nefes_hip.h"

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

}  // namespace

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
