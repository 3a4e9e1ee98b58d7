// 64-lane wavefront helpers for the one-wave-per-ray kernels (gfx950: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

// Sum over the 64 lanes, result on every lane.  Within a row of 16 lanes the exchange runs on DPP (register-to-register,
// no LDS crossbar: a ds_bpermute butterfly costs a round trip per step and the one-wave-per-ray kernels issue dozens of sums
// back to back); the four row sums are then read with v_readlane and added in lane order.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xb1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4e>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror: every lane now holds its row's sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return t;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix product over lanes 0..63
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
// The same scan on DPP (register to register: no LDS crossbar round trip per step): row_shr 1, 2, 4, 8 inside the rows of 16
// lanes -- a lane that has no source keeps its value, the shifted-in operand reads as zero -- then the rows' totals are added
// across rows (v_readlane of lanes 15, 31, 47).  Adds the same terms in a different association than the ds_bpermute form.
template <int CTRL>
__device__ __forceinline__ double dpp_shr_zero(double v) {      // lane i <- lane i - k of its row, 0 where there is none
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_sum_dpp(double v, int lane) {
    v += dpp_shr_zero<0x111>(v);     // row_shr:1
    v += dpp_shr_zero<0x112>(v);     // row_shr:2
    v += dpp_shr_zero<0x114>(v);     // row_shr:4
    v += dpp_shr_zero<0x118>(v);     // row_shr:8
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const int row = lane >> 4;
    const double base = row == 0 ? 0.0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : (r0 + r1) + r2));
    return base + v;
}
// Reverse scan of affine maps f_i(x) = a_i + m_i * x over lanes: on return lane i holds the
// composition F_i = f_i o f_{i+1} o ... o f_63  as (m, a).
__device__ __forceinline__ void wave_rev_affine(double& m, double& a, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double m2 = __shfl_down(m, o), a2 = __shfl_down(a, o);
        if (lane + o < 64) {   // F_i <- F_i o F_{i+o}
            a = a + m * a2;
            m = m * m2;
        }
    }
}
__device__ __forceinline__ double lane_bcast(double v, int src) { return __shfl(v, src); }

// ---- rays that occupy RW rows of 16 lanes, four consecutive samples per lane (composite.hip's four-samples-per-lane kernels and
// ---- the fused coarse-pass sampler of sample_pdf.hip): 4 / RW rays share a wave -----------------------------------------------
template <int RW> struct Seg {
    static constexpr int LPR = 16 * RW;              // lanes per ray
    static constexpr int RPW = 4 / RW;               // rays per wave (RW = 3: one ray, the fourth row idles)
};
// sum over the lanes of this lane's ray (RW rows of 16 lanes), on every lane of the ray
template <int RW>
__device__ __forceinline__ double seg_sum(double v, int lane) {
    v += dpp_move<0xb1>(v);
    v += dpp_move<0x4e>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);                         // every lane holds its row's sum
    if (RW == 1) return v;
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    if (RW == 2) return lane < 32 ? r0 + r1 : r2 + r3;
    if (RW == 3) return (r0 + r1) + r2;
    return ((r0 + r1) + r2) + r3;
}
// inclusive prefix product over the lanes of a ray (sl = lane index inside the ray)
template <int RW>
__device__ __forceinline__ double seg_incl_prod(double v, int sl) {
#pragma unroll
    for (int o = 1; o < 16 * RW; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (sl >= o) v *= u;
    }
    return v;
}
// reverse scan of affine maps over the lanes of a ray: lane i ends with F_i o F_{i+1} o ... o F_last
template <int RW>
__device__ __forceinline__ void seg_rev_affine(double& m, double& a, int sl) {
#pragma unroll
    for (int o = 1; o < 16 * RW; o <<= 1) {
        const double m2 = __shfl_down(m, o), a2 = __shfl_down(a, o);
        if (sl + o < 16 * RW) {
            a = a + m * a2;
            m = m * m2;
        }
    }
}
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float el(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
// inclusive prefix sum over the lanes of a ray (sl = lane index inside the ray)
template <int RW>
__device__ __forceinline__ double seg_incl_sum(double v, int sl) {
#pragma unroll
    for (int o = 1; o < 16 * RW; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (sl >= o) v += u;
    }
    return v;
}

// ---- the same scans for a ray that occupies exactly ONE row of 16 lanes (RW = 1), on DPP row shifts: register to register, no
// ---- ds_bpermute round trips (the fused coarse-pass sampler is bound by LDS instructions).  Same Hillis-Steele tree, same results.
template <int CTRL>
__device__ __forceinline__ double dpp_shr_keep(double v, double keep) {      // lane i <- lane i - k of its row; `keep` where there is none
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(keep), lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_incl_prod(double v) {
    v *= dpp_shr_keep<0x111>(v, 1.0);
    v *= dpp_shr_keep<0x112>(v, 1.0);
    v *= dpp_shr_keep<0x114>(v, 1.0);
    v *= dpp_shr_keep<0x118>(v, 1.0);
    return v;
}
__device__ __forceinline__ double row_incl_sum(double v) {
    v += dpp_shr_zero<0x111>(v);
    v += dpp_shr_zero<0x112>(v);
    v += dpp_shr_zero<0x114>(v);
    v += dpp_shr_zero<0x118>(v);
    return v;
}
__device__ __forceinline__ int row_incl_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    return v;
}
// previous / next lane of the row (`keep` at the row's first / last lane)
__device__ __forceinline__ double row_prev(double v, double keep) { return dpp_shr_keep<0x111>(v, keep); }
__device__ __forceinline__ float row_next(float v) {             // row_shl:1; the row's last lane keeps its own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x101, 0xf, 0xf, false));
}
