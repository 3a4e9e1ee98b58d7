// Ray generation / NDC warp / coarse depths and the pose-gradient reductions.
// Restates script/models/ray_utils.py:5-16 (get_rays), :27-44 (ndc_rays), rendering.py:96-112 (z_vals),
// :217 (viewdirs) and the autograd backward of those ops down to the 3x4 camera-to-world pose.
// All HBM-bound and tiny next to the MLP (36 B written per ray forward, 36 B read per ray backward).
#include "../../include/nefes_hip.h"
#include "wave.h"

__device__ __forceinline__ void pixel_dir(int H, int W, float focal, int row, int col, float (&dc)[3]) {
    // dirs = [(i - W/2)/f, -(j - H/2)/f, -1]   (ray_utils.py:10; no half-pixel offset)
    dc[0] = __fdiv_rn(__fsub_rn((float)col, (float)(W * .5)), focal);
    dc[1] = -__fdiv_rn(__fsub_rn((float)row, (float)(H * .5)), focal);
    dc[2] = -1.f;
}

__global__ __launch_bounds__(256) void raygen_fwd_kernel(int H, int W, float focal, const float* __restrict__ c2w, int row0,
                                                         int n, float* rays_o, float* rays_d, float* viewdirs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int row = row0 + idx / W, col = idx % W;
    float dc[3];
    pixel_dir(H, W, focal, row, col, dc);
    float d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // rays_d[a] = sum_b dirs[b] * c2w[a][b]  (:13)
        const float p0 = __fmul_rn(dc[0], c2w[a * 4 + 0]), p1 = __fmul_rn(dc[1], c2w[a * 4 + 1]),
                    p2 = __fmul_rn(dc[2], c2w[a * 4 + 2]);
        d[a] = __fadd_rn(__fadd_rn(p0, p1), p2);
    }
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rays_o[(size_t)idx * 3 + a] = c2w[a * 4 + 3];                    // :15
        rays_d[(size_t)idx * 3 + a] = d[a];
        if (viewdirs) viewdirs[(size_t)idx * 3 + a] = __fdiv_rn(d[a], nrm);   // rendering.py:217
    }
}

// stage 1: per-block partial of the 12 pose-gradient sums (f64), stage 2: fixed-order final sum.
__global__ __launch_bounds__(256) void raygen_bwd_kernel(int H, int W, float focal, const float* __restrict__ c2w, int row0,
                                                         int n, const float* g_o, const float* g_d, const float* g_v,
                                                         double* partial) {
    __shared__ double red[4][12];
    double s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0.0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int row = row0 + idx / W, col = idx % W;
        float dc[3];
        pixel_dir(H, W, focal, row, col, dc);
        float gd[3] = {0.f, 0.f, 0.f};
        if (g_d)
#pragma unroll
            for (int a = 0; a < 3; ++a) gd[a] = g_d[(size_t)idx * 3 + a];
        if (g_v) {   // v = d/|d|  =>  d L/d d += (g - v (v.g)) / |d|
            float d[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                d[a] = __fadd_rn(__fadd_rn(__fmul_rn(dc[0], c2w[a * 4]), __fmul_rn(dc[1], c2w[a * 4 + 1])), __fmul_rn(dc[2], c2w[a * 4 + 2]));
            const float nrm = __fsqrt_rn(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float gv0 = g_v[(size_t)idx * 3], gv1 = g_v[(size_t)idx * 3 + 1], gv2 = g_v[(size_t)idx * 3 + 2];
            const float v0 = d[0] / nrm, v1 = d[1] / nrm, v2 = d[2] / nrm;
            const float dot = v0 * gv0 + v1 * gv1 + v2 * gv2;
            gd[0] += (gv0 - v0 * dot) / nrm;
            gd[1] += (gv1 - v1 * dot) / nrm;
            gd[2] += (gv2 - v2 * dot) / nrm;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) s[a * 4 + b] += (double)gd[a] * (double)dc[b];
            if (g_o) s[a * 4 + 3] += (double)g_o[(size_t)idx * 3 + a];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const double v = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) partial[(size_t)blockIdx.x * 12 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ void raygen_bwd_final(const double* partial, int nblocks, float* g_c2w) {
    const int k = threadIdx.x;
    if (k >= 12) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * 12 + k];
    g_c2w[k] = (float)s;
}

static int raygen_blocks(int n) {
    int b = (n + 255) / 256;
    return b > 1024 ? 1024 : (b < 1 ? 1 : b);
}

extern "C" int nefes_raygen_fwd(int H, int W, float focal, const float* c2w, int row0, int nrows, float* rays_o,
                                float* rays_d, float* viewdirs, void* stream) {
    if (!c2w || !rays_o || !rays_d || H <= 0 || W <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > H) return NEFES_E_BADARG;
    const int n = nrows * W;
    hipLaunchKernelGGL(raygen_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, c2w, row0, n,
                       rays_o, rays_d, viewdirs);
    return (int)hipGetLastError();
}
extern "C" size_t nefes_raygen_bwd_workspace(int n_rays) { return (size_t)raygen_blocks(n_rays) * 12 * sizeof(double); }
extern "C" int nefes_raygen_bwd(int H, int W, float focal, const float* c2w, int row0, int nrows, const float* g_rays_o,
                                const float* g_rays_d, const float* g_viewdirs, void* workspace, float* g_c2w,
                                void* stream) {
    if (!c2w || !workspace || !g_c2w || H <= 0 || W <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > H) return NEFES_E_BADARG;
    const int n = nrows * W, nb = raygen_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(raygen_bwd_kernel, dim3(nb), dim3(256), 0, st, H, W, focal, c2w, row0, n, g_rays_o, g_rays_d,
                       g_viewdirs, (double*)workspace);
    hipLaunchKernelGGL(raygen_bwd_final, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, g_c2w);
    return (int)hipGetLastError();
}

// ---- ndc_rays (ray_utils.py:27-44) ------------------------------------------------------------
struct Ndc {
    float t, ox, oy, oz, kx, ky;
};
__device__ __forceinline__ Ndc ndc_common(int H, int W, float focal, float near, const float* o, const float* d) {
    Ndc r;
    r.t = -(near + o[2]) / d[2];
    r.ox = o[0] + r.t * d[0];
    r.oy = o[1] + r.t * d[1];
    r.oz = o[2] + r.t * d[2];
    r.kx = (float)(-1. / (W / (2. * (double)focal)));
    r.ky = (float)(-1. / (H / (2. * (double)focal)));
    return r;
}
__global__ __launch_bounds__(256) void ndc_fwd_kernel(int H, int W, float focal, float near, int n, const float* ro,
                                                      const float* rd, float* oo, float* od) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* o = ro + (size_t)i * 3;
    const float* d = rd + (size_t)i * 3;
    const Ndc r = ndc_common(H, W, focal, near, o, d);
    oo[(size_t)i * 3 + 0] = r.kx * r.ox / r.oz;
    oo[(size_t)i * 3 + 1] = r.ky * r.oy / r.oz;
    oo[(size_t)i * 3 + 2] = 1.f + 2.f * near / r.oz;
    od[(size_t)i * 3 + 0] = r.kx * (d[0] / d[2] - r.ox / r.oz);
    od[(size_t)i * 3 + 1] = r.ky * (d[1] / d[2] - r.oy / r.oz);
    od[(size_t)i * 3 + 2] = -2.f * near / r.oz;
}
__global__ __launch_bounds__(256) void ndc_bwd_kernel(int H, int W, float focal, float near, int n, const float* ro,
                                                      const float* rd, const float* goo, const float* god, float* gro,
                                                      float* grd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* o = ro + (size_t)i * 3;
    const float* d = rd + (size_t)i * 3;
    const Ndc r = ndc_common(H, W, focal, near, o, d);
    float go[3] = {0, 0, 0}, gd[3] = {0, 0, 0};
    if (goo) { go[0] = goo[(size_t)i * 3]; go[1] = goo[(size_t)i * 3 + 1]; go[2] = goo[(size_t)i * 3 + 2]; }
    if (god) { gd[0] = god[(size_t)i * 3]; gd[1] = god[(size_t)i * 3 + 1]; gd[2] = god[(size_t)i * 3 + 2]; }
    const float iz = 1.f / r.oz, idz = 1.f / d[2];
    // gradients w.r.t. the shifted origin (ox,oy,oz) and the direction terms d0/d2, d1/d2
    const float g_ox = (go[0] - gd[0]) * r.kx * iz;
    const float g_oy = (go[1] - gd[1]) * r.ky * iz;
    const float g_oz = -(go[0] - gd[0]) * r.kx * r.ox * iz * iz - (go[1] - gd[1]) * r.ky * r.oy * iz * iz
                       - (go[2] - gd[2]) * 2.f * near * iz * iz;
    float g_d0 = gd[0] * r.kx * idz, g_d1 = gd[1] * r.ky * idz;
    float g_d2 = -gd[0] * r.kx * d[0] * idz * idz - gd[1] * r.ky * d[1] * idz * idz;
    // shifted origin = o + t d, t = -(near + o2)/d2
    const float g_t = g_ox * d[0] + g_oy * d[1] + g_oz * d[2];
    g_d0 += g_ox * r.t; g_d1 += g_oy * r.t; g_d2 += g_oz * r.t;
    const float g_o2 = g_oz + g_t * (-idz);
    g_d2 += g_t * ((near + o[2]) * idz * idz);
    gro[(size_t)i * 3 + 0] = g_ox; gro[(size_t)i * 3 + 1] = g_oy; gro[(size_t)i * 3 + 2] = g_o2;
    grd[(size_t)i * 3 + 0] = g_d0; grd[(size_t)i * 3 + 1] = g_d1; grd[(size_t)i * 3 + 2] = g_d2;
}
extern "C" int nefes_ndc_fwd(int H, int W, float focal, float near, int n, const float* rays_o, const float* rays_d,
                             float* out_o, float* out_d, void* stream) {
    if (!rays_o || !rays_d || !out_o || !out_d || n <= 0) return NEFES_E_BADARG;
    hipLaunchKernelGGL(ndc_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, near, n, rays_o,
                       rays_d, out_o, out_d);
    return (int)hipGetLastError();
}
extern "C" int nefes_ndc_bwd(int H, int W, float focal, float near, int n, const float* rays_o, const float* rays_d,
                             const float* g_out_o, const float* g_out_d, float* g_rays_o, float* g_rays_d, void* stream) {
    if (!rays_o || !rays_d || !g_rays_o || !g_rays_d || n <= 0) return NEFES_E_BADARG;
    hipLaunchKernelGGL(ndc_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, near, n, rays_o,
                       rays_d, g_out_o, g_out_d, g_rays_o, g_rays_d);
    return (int)hipGetLastError();
}

// ---- coarse depths (rendering.py:96-112) ---------------------------------------------------------
// bounds != null: near/far are read per ray from bounds[ray*stride + 0/1] (the [n,21] ray batch of rendering.py:227-235,
// columns 6:8), as render_rays does (:90-100); otherwise the scalars apply to every ray.
__global__ __launch_bounds__(256) void coarse_depths_kernel(int N, int Nc, float near, float far, const float* __restrict__ bounds,
                                                            int stride, int lindisp, const float* __restrict__ t,
                                                            const float* t_rand, float* z) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * Nc) return;
    const int k = (int)(i % Nc);
    if (bounds) {
        const size_t ray = i / Nc;
        near = bounds[ray * stride];
        far = bounds[ray * stride + 1];
    }
    auto zk = [&](int kk) {
        const float tt = t[kk];
        if (!lindisp) return __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, tt)), __fmul_rn(far, tt));
        return __fdiv_rn(1.f, __fadd_rn(__fmul_rn(__fdiv_rn(1.f, near), __fsub_rn(1.f, tt)), __fmul_rn(__fdiv_rn(1.f, far), tt)));
    };
    float v = zk(k);
    if (t_rand) {   // stratified jitter inside [lower, upper] (:104-112)
        const float lo = k == 0 ? v : __fmul_rn(.5f, __fadd_rn(v, zk(k - 1)));
        const float hi = k == Nc - 1 ? v : __fmul_rn(.5f, __fadd_rn(zk(k + 1), v));
        v = __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t_rand[i]));
    }
    z[i] = v;
}
extern "C" int nefes_coarse_depths(int N, int Nc, float near, float far, int lindisp, const float* t, const float* t_rand,
                                   float* z, void* stream) {
    if (!t || !z || N <= 0 || Nc <= 0) return NEFES_E_BADARG;
    const size_t n = (size_t)N * Nc;
    hipLaunchKernelGGL(coarse_depths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, Nc, near,
                       far, (const float*)nullptr, 0, lindisp, t, t_rand, z);
    return (int)hipGetLastError();
}
extern "C" int nefes_coarse_depths_rays(int N, int Nc, const float* bounds, int stride, int lindisp, const float* t,
                                        const float* t_rand, float* z, void* stream) {
    if (!t || !z || !bounds || stride < 2 || N <= 0 || Nc <= 0) return NEFES_E_BADARG;
    const size_t n = (size_t)N * Nc;
    hipLaunchKernelGGL(coarse_depths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, Nc, 0.f,
                       0.f, bounds, stride, lindisp, t, t_rand, z);
    return (int)hipGetLastError();
}

// ---- per-ray reduction of the field backward (autograd of pts = o + d*z, rendering.py:142) ----------
__global__ __launch_bounds__(256) void ray_grad_reduce_kernel(int N, int S, const float* __restrict__ z,
                                                              const float* __restrict__ g_pts, const float* __restrict__ g_vs,
                                                              float* g_o, float* g_d, float* g_v) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float* gp = g_pts + (size_t)ray * S * 3;
    const float* gv = g_vs ? g_vs + (size_t)ray * S * 3 : nullptr;
    const float* zr = z + (size_t)ray * S;
    double so[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
    for (int i = lane; i < 3 * S; i += 64) {
        const int s = i / 3, c = i - 3 * s;
        const float g = gp[i];
        const double gz = (double)(g * zr[s]);
        const float v = gv ? gv[i] : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (c == k) { so[k] += g; sd[k] += gz; sv[k] += v; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = wave_sum(so[k]), b = wave_sum(sd[k]), c = wave_sum(sv[k]);
        if (lane == 0) {
            g_o[(size_t)ray * 3 + k] = (float)a;
            g_d[(size_t)ray * 3 + k] = (float)b;
            if (g_v) g_v[(size_t)ray * 3 + k] = (float)c;
        }
    }
}
extern "C" int nefes_ray_grad_reduce(int N, int S, const float* z, const float* g_pts, const float* g_viewdirs_s,
                                     float* g_rays_o, float* g_rays_d, float* g_viewdirs, void* stream) {
    if (!z || !g_pts || !g_rays_o || !g_rays_d || N <= 0 || S <= 0) return NEFES_E_BADARG;
    hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, S, z, g_pts,
                       g_viewdirs_s, g_rays_o, g_rays_d, g_viewdirs);
    return (int)hipGetLastError();
}
