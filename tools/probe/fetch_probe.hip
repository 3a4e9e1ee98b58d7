// Calibration of the FETCH_SIZE counter for the two load widths the field kernels use: a buffer of known size is read once with
// 4-byte-per-lane loads (the backward's ReLU-mask words) and once with 16-byte-per-lane loads; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_probe
// and compare the counter (KiB) x 1024 with the bytes printed here.   hipcc --offload-arch=gfx950 -O2 tools/probe/fetch_probe.hip -o /tmp/fp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void read_dword_kernel(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_dwordx4_kernel(const u32x4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    uint32_t *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    read_dword_kernel<<<256 * 8, 256>>>(buf, bytes / 4, out);
    read_dwordx4_kernel<<<256 * 8, 256>>>((const u32x4*)buf, bytes / 16, out);
    (void)hipDeviceSynchronize();
    printf("each kernel read %zu bytes (%.3f GB)\n", bytes, bytes / 1e9);
    return 0;
}
