// Store-pattern probe for the train-mode activation buffers: a persistent grid (one 256-thread workgroup per CU, as the
// TRAIN field kernels run) writes `rows` x 128-sample tiles, non-temporally, in three patterns:
//   A  what train_save_h3 / StoringSplitH issue: one float per lane and instruction, a wave's instruction = 4 segments of 64 bytes
//      (blocks of 32 rows x 16 samples, rows rho(0,r) and +4 for the upper lane half)
//   B  the same bytes as 16 bytes per lane and instruction, a wave's instruction = 1 KiB contiguous
//   C  as A but 8 bytes per lane (two neighbouring samples per lane: what a lane-pair exchange would allow)
// hipcc --offload-arch=gfx950 -O2 tools/probe/store_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ int rho(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(float* buf, int n_tiles, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        float* t = buf + (size_t)tile * rows * 128;
        const float v = (float)(tile + lane);
        for (int blk = 0; blk < rows / 32; ++blk) {
            if (PAT == 0) {
                float* p = t + (size_t)blk * 4096 + (((wave * 32 + j) >> 4) * 512 + 4 * h * 16 + (j & 15));
#pragma unroll
                for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v + r, &p[rho(0, r) * 16]);
            } else if (PAT == 1) {
                f4* p = (f4*)(t + (size_t)blk * 4096 + wave * 1024) + lane;          // 4 x (64 lanes x 16 B)
#pragma unroll
                for (int r = 0; r < 4; ++r) { f4 x = {v + r, v, v, v}; __builtin_nontemporal_store(x, &p[r * 64]); }
            } else if (PAT == 3) {
                // D: blocks of 32 rows x 32 samples (4 KiB): one instruction = two whole 128-byte lines (rows rho(0,r) and +4)
                float* p = t + (size_t)blk * 4096 + wave * 1024 + 4 * h * 32 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v + r, &p[rho(0, r) * 32]);
            } else if (PAT == 4) {
                // E: as A, ordinary (write-back) stores
                float* p = t + (size_t)blk * 4096 + (((wave * 32 + j) >> 4) * 512 + 4 * h * 16 + (j & 15));
#pragma unroll
                for (int r = 0; r < 16; ++r) p[rho(0, r) * 16] = v + r;
            } else if (PAT == 5) {
                // F: the layout of A, registers r and r+1 exchanged across the lane halves first (v_permlane32_swap): lane (j, h) then
                // holds rows rho(0,r) + h and rho(0,r) + 4 + h, so one instruction = two whole 128-byte lines (rows 2k, 2k+1 of two blocks)
                float* p = t + (size_t)blk * 4096 + (((wave * 32 + j) >> 4) * 512 + h * 16 + (j & 15));
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v + r), __float_as_uint(v + r + 1), false, false);
                    __builtin_nontemporal_store(__uint_as_float(sw[0]), &p[rho(0, r) * 16]);
                    __builtin_nontemporal_store(__uint_as_float(sw[1]), &p[(rho(0, r) + 4) * 16]);
                }
            } else {
                // 8 bytes per lane: lane (j', h) owns samples 2j', 2j'+1 of 8 rows
                f2* p = (f2*)(t + (size_t)blk * 4096 + (((wave * 32 + 2 * (j & 15)) >> 4) * 512 + (j & 7) * 2)) ;
                const int rbase = (j >> 4) * 2 + 4 * h;
#pragma unroll
                for (int r = 0; r < 8; ++r) { f2 x = {v + r, v}; __builtin_nontemporal_store(x, &p[(((r >> 1) * 8 + rbase + (r & 1)) * 16) / 2]); }
            }
        }
    }
}

int main() {
    const int rows = 1696, n_tiles = 20000;                       // ~ the configs[0] step: 2.56 M samples x 6.9 KB
    const size_t bytes = (size_t)n_tiles * rows * 128 * 4;
    float* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid = 256; grid <= 512; grid += 256)
    for (int pat = 0; pat < 6; ++pat)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) store_kernel<0><<<grid, 256>>>(buf, n_tiles, rows);
            if (pat == 1) store_kernel<1><<<grid, 256>>>(buf, n_tiles, rows);
            if (pat == 2) store_kernel<2><<<grid, 256>>>(buf, n_tiles, rows);
            if (pat == 3) store_kernel<3><<<grid, 256>>>(buf, n_tiles, rows);
            if (pat == 4) store_kernel<4><<<grid, 256>>>(buf, n_tiles, rows);
            if (pat == 5) store_kernel<5><<<grid, 256>>>(buf, n_tiles, rows);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("grid %d pattern %c: %.3f ms  %.2f TB/s  (%.2f GB)\n", grid, 'A' + pat, ms, bytes / ms / 1e9, bytes / 1e9);
        }
    return 0;
}
