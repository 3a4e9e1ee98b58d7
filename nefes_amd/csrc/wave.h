// 64-lane wavefront helpers for the one-wave-per-ray kernels (gfx950: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
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
