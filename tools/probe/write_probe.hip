// HBM write rate of the train-mode kernels' store PATTERN, in isolation (round 6).  The TRAIN forward / dX
// kernels write 13-14 GB per step at 3.7-3.9 TB/s; the compositor's backward writes at 5.7.  Is it the pattern?  Every wave writes
// tiles of 32 rows x 32 samples of fp32 (4 KiB) the way a wave of the field kernels holds them -- lane (j, h): sample j, rows
// rho(r) + 4 h for its sixteen registers r -- in four layouts:
//   0  the shipped one: [row/32][sample/16][row%32][sample%16]: a store instruction = four 64-byte segments (global_store_dword nt)
//   1  [row/32][sample/32][row%32][sample%32]: a store instruction = two full 128-byte lines (global_store_dword nt)
//   2  fully contiguous 16 bytes per lane (global_store_dwordx4 nt): what an LDS transpose in front of the stores would give
//   3  layout 0 with ordinary (write-back) stores
//     hipcc --offload-arch=gfx950 -O3 -o tools/probe/write_probe tools/probe/write_probe.hip && tools/probe/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int rho(int r) { return 8 * (r >> 2) + (r & 3); }

template <int MODE>
__global__ __launch_bounds__(256) void write_kernel(float* __restrict__ out, long tiles_per_wave, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long wave_id = (long)blockIdx.x * 4 + wave;
    float* base = out + wave_id * tiles_per_wave * 1024;          // 32 rows x 32 samples per tile
    for (long t = 0; t < tiles_per_wave; ++t) {
        float* p = base + t * 1024;
        if (MODE == 0 || MODE == 3) {
            float* q = p + (j >> 4) * 512 + 4 * h * 16 + (j & 15);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (MODE == 0) __builtin_nontemporal_store(v + (float)r, &q[rho(r) * 16]);
                else q[rho(r) * 16] = v + (float)r;
            }
        } else if (MODE == 1) {
            float* q = p + 4 * h * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v + (float)r, &q[rho(r) * 32]);
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4* q = (f4*)p + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f4 x = {v, v + 1.f, v + 2.f, v + (float)r};
                __builtin_nontemporal_store(x, &q[r * 64]);
            }
        }
    }
}

int main() {
    const long bytes = 12L << 30;                                  // ~ one train-mode forward's activation buffer
    float* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const int blocks = 512;                                        // two workgroups per CU, as the Wd = 128 kernels run
    const long tiles_per_wave = bytes / 4096 / (blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"shipped layout, 4 x 64 B per store instruction, nt", "2 x 128 B lines per store instruction, nt",
                            "16 B per lane contiguous (dwordx4), nt", "shipped layout, write-back stores"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 4; ++m) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(write_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, tiles_per_wave, 1.f);
            if (m == 1) hipLaunchKernelGGL(write_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, tiles_per_wave, 1.f);
            if (m == 2) hipLaunchKernelGGL(write_kernel<2>, dim3(blocks), dim3(256), 0, 0, d, tiles_per_wave, 1.f);
            if (m == 3) hipLaunchKernelGGL(write_kernel<3>, dim3(blocks), dim3(256), 0, 0, d, tiles_per_wave, 1.f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double gb = (double)tiles_per_wave * blocks * 4 * 4096 / 1e9;
            if (rep) printf("%-58s %7.3f ms  %6.2f TB/s\n", names[m], ms, gb / ms);
        }
    hipFree(d);
    return 0;
}
