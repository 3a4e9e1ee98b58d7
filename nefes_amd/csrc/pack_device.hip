// Device-side weight repack: blob[slot] = part(flat[(code >> 3) - 1], code & 7) for every 16-bit slot after the header, with the
// slot -> code map of nefes_pack_map (pack.cpp).  One launch re-packs every stream of a network from the concatenated
// parameter vector, so a training step (script/run_nefes.py:42-108: optimizer.step() changes the weights every iteration)
// never copies parameters to the host.  Bit-identical to nefes_pack_weights (tests/test_gpu_train.py).
// The fp16 two-part streams (field_h3.h) store w * 2^e with one exponent e per weight matrix and carry a table of row bounds
// and bias maxima: a first launch (one workgroup per job of nefes_pack_h3_plan) computes those from the same parameter
// vector -- the exponents into a scratch array, bounds and maxima straight into the blob -- then the expansion reads the
// exponents (parts 5 / 6 = fp16 hi / lo, part 7 = the exponent word itself).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/nefes_hip.h"
#include "layout.h"

namespace {

__device__ __forceinline__ uint32_t rne_bf16(float f) {
    uint32_t b = __float_as_uint(f);
    b += 0x7fffu + ((b >> 16) & 1u);
    return b >> 16;
}

__device__ __forceinline__ uint32_t f16_bits(float y) { return (uint32_t)__half_as_ushort(__float2half_rn(y)); }

__device__ __forceinline__ uint32_t part16(const float* flat, const int* gexp, uint32_t code) {
    if (code == 0u) return 0u;
    const uint32_t part = code & 7u, field = (code >> 3) & 0xffffffu, group = code >> 27;
    if (part == 7u) {                                    // exponent word of a group: field 1 = low half, 2 = high half
        const uint32_t e = (uint32_t)gexp[group];
        return field == 1u ? (e & 0xffffu) : (e >> 16);
    }
    const float x = flat[field - 1u];
    const uint32_t b = __float_as_uint(x);
    if (part == 0u) return b & 0xffffu;
    if (part == 1u) return b >> 16;
    if (part >= 5u) {                                    // fp16 two-part split of x 2^e, as split_f16x2 (pack.cpp)
        const float y = ldexpf(x, gexp[group]);
        const uint32_t hi = f16_bits(y);
        if (part == 5u) return hi;
        return f16_bits(__fsub_rn(y, __half2float(__ushort_as_half((unsigned short)hi))));
    }
    // bf16 triple, as split_bf16x3 (pack.cpp): round-to-nearest-even parts, both subtractions exact
    const uint32_t hi = rne_bf16(x);
    if (part == 2u) return hi;
    const float r = __fsub_rn(x, __uint_as_float(hi << 16));
    const uint32_t mid = rne_bf16(r);
    if (part == 3u) return mid;
    return rne_bf16(__fsub_rn(r, __uint_as_float(mid << 16)));
}

__global__ __launch_bounds__(256) void pack_device_kernel(const float* __restrict__ flat, const uint2* __restrict__ map2,
                                                          const int* __restrict__ gexp, long long first_word, long long n_words,
                                                          uint32_t* __restrict__ blob) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long w = first_word + (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint2 c = map2[w];
        if (c.x == NEFES_PACK_KEEP) continue;                       // written by h3_scales_kernel (or, without a plan, left as it is)
        if (!gexp && ((c.x & 7u) >= 5u || (c.y & 7u) >= 5u)) continue;   // no plan: the fp16 streams stay as the host packer wrote them
        blob[w] = part16(flat, gexp, c.x) | (part16(flat, gexp, c.y) << 16);
    }
}

// exponent e with max|w| * 2^e in [2^NEFES_H3_TARGET_EXP, 2^(NEFES_H3_TARGET_EXP+1)) -- scale_exp of pack.cpp
__device__ __forceinline__ int scale_exp_dev(float amax) {
    if (!(amax > 0.f) || !isfinite(amax)) return 0;
    int e;
    (void)frexpf(amax, &e);
    const int r = NEFES_H3_TARGET_EXP + 1 - e;
    return r < -60 ? -60 : (r > 60 ? 60 : r);
}

// one workgroup per job of the plan (pack.cpp nefes_pack_h3_plan)
__global__ __launch_bounds__(256) void h3_scales_kernel(const float* __restrict__ flat, const int* __restrict__ plan,
                                                        int* __restrict__ gexp, uint32_t* __restrict__ blob) {
    __shared__ float red[256];
    const int* j = plan + 2 + 8 * blockIdx.x;
    const int kind = j[0], out = j[1], tid = threadIdx.x;
    float best = 0.f;
    if (kind == 1) {     // max over rows of sum_k |w|: the sum in double and in the host's order (Seg::row_bound), so the bound is bit-identical
        const int n_rows = j[2], n_k = j[3];
        const int* A = plan + j[4];
        const int* B = plan + j[5];
        for (int r = tid; r < n_rows; r += 256) {
            double sum = 0.0;
            const int a = A[r];
            for (int k = 0; k < n_k; ++k) sum += fabs((double)flat[a + B[k]]);
            best = fmaxf(best, (float)sum);
        }
    } else {
        for (int q = 0; q < 3; ++q) {
            const int off = j[2 + 2 * q], cnt = j[3 + 2 * q];
            for (int i = tid; i < cnt; i += 256) best = fmaxf(best, fabsf(flat[off + i]));
        }
    }
    red[tid] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) {
        const float m = red[0];
        if (kind == 0) gexp[out] = scale_exp_dev(m);
        else if (kind == 1) blob[out] = __float_as_uint(__fmul_rn(m, 1.0001f));
        else blob[out] = __float_as_uint(m);
    }
}

}  // namespace

extern "C" int nefes_pack_device(const float* flat, int64_t n_params, const uint32_t* map, int64_t n_entries, const int32_t* plan,
                                 int n_jobs, int32_t* scratch, void* blob, void* stream) {
    if (!flat || !map || !blob || n_params <= 0 || n_entries <= NEFES_BLOB_HEADER_BYTES / 2 || (n_entries & 1)) return NEFES_E_BADARG;
    if (plan && (!scratch || n_jobs <= 0)) return NEFES_E_BADARG;
    if (plan) hipLaunchKernelGGL(h3_scales_kernel, dim3((unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, flat, plan, scratch, (uint32_t*)blob);
    const long long n_words = n_entries / 2, first = NEFES_BLOB_HEADER_BYTES / 4;     // the header stays as the host packer wrote it
    long long blocks = (n_words - first + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pack_device_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, (const uint2*)map,
                       plan ? (const int*)scratch : (const int*)nullptr, first, n_words, (uint32_t*)blob);
    return (int)hipGetLastError();
}
