// Device-side weight repack: blob[slot] = part(flat[(code >> 3) - 1], code & 7) for every 16-bit slot after the header, with the
// slot -> code map of nefes_pack_map (pack.cpp).  One launch re-packs every stream of a network from the concatenated
// parameter vector, so a training step (script/run_nefes.py:42-108: optimizer.step() changes the weights every iteration)
// never copies parameters to the host.  Bit-identical to nefes_pack_weights (tests/test_gpu_train.py).
#include <hip/hip_runtime.h>

#include "../../include/nefes_hip.h"
#include "layout.h"

namespace {

__device__ __forceinline__ uint32_t rne_bf16(float f) {
    uint32_t b = __float_as_uint(f);
    b += 0x7fffu + ((b >> 16) & 1u);
    return b >> 16;
}

__device__ __forceinline__ uint32_t part16(const float* flat, uint32_t code) {
    if (code == 0u) return 0u;
    const float x = flat[(code >> 3) - 1u];
    const uint32_t part = code & 7u, b = __float_as_uint(x);
    if (part == 0u) return b & 0xffffu;
    if (part == 1u) return b >> 16;
    // bf16 triple, as split_bf16x3 (pack.cpp): round-to-nearest-even parts, both subtractions exact
    const uint32_t hi = rne_bf16(x);
    if (part == 2u) return hi;
    const float r = __fsub_rn(x, __uint_as_float(hi << 16));
    const uint32_t mid = rne_bf16(r);
    if (part == 3u) return mid;
    return rne_bf16(__fsub_rn(r, __uint_as_float(mid << 16)));
}

__global__ __launch_bounds__(256) void pack_device_kernel(const float* __restrict__ flat, const uint2* __restrict__ map2,
                                                          long long first_word, long long n_words, uint32_t* __restrict__ blob) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long w = first_word + (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint2 c = map2[w];
        blob[w] = part16(flat, c.x) | (part16(flat, c.y) << 16);
    }
}

}  // namespace

extern "C" int nefes_pack_device(const float* flat, int64_t n_params, const uint32_t* map, int64_t n_entries, void* blob,
                                 void* stream) {
    if (!flat || !map || !blob || n_params <= 0 || n_entries <= NEFES_BLOB_HEADER_BYTES / 2 || (n_entries & 1)) return NEFES_E_BADARG;
    const long long n_words = n_entries / 2, first = NEFES_BLOB_HEADER_BYTES / 4;     // the header stays as the host packer wrote it
    long long blocks = (n_words - first + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pack_device_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, (const uint2*)map, first,
                       n_words, (uint32_t*)blob);
    return (int)hipGetLastError();
}
