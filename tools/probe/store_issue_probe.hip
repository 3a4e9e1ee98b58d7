// What a store costs INSIDE a busy wave: the TRAIN field kernels issue one 4-byte-per-lane global_store_dword per MFMA
// next to ~6 vector instructions per MFMA, two workgroups per CU.  Removing those stores takes a quarter off the kernels, sending them
// to an L2-resident buffer does not (profiles/r04/secondary/ab_train_stores.log) -- so it is not the memory system.  This probe runs the
// same regime (8 waves per CU, per iteration 4 MFMAs + 24 v_fma_f32 + the stores of 16 bytes per lane) and varies only how those
// 16 bytes per lane leave:
//     0  not at all                         1  four global_store_dword (one per MFMA), lane-contiguous 256-byte rows, as the kernels
//     2  two global_store_dwordx2           3  one global_store_dwordx4
//     4  four global_store_dword through the saddr form (scalar base + 32-bit lane offset), written as asm
// All stores non-temporal, into a buffer far larger than the caches (the HBM write rate then bounds every variant alike) and into 8 KiB
// per wave that stay in L2 (what is left is the cost of issuing them).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/store_issue_probe.hip -o tools/probe/store_issue_probe && tools/probe/store_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool RESIDENT, int DIST>
__global__ __launch_bounds__(256, 2) void busy(float* buf, size_t floats_per_wave, int iters, float* sink, const f32x4* stream) {
    const int lane = threadIdx.x & 63;
    const size_t wave_id = (size_t)blockIdx.x * 4 + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* base = buf + wave_id * floats_per_wave;              // wave-uniform
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.01f * ((lane * 7 + i) % 13)); b[i] = (_Float16)(0.25f + 0.01f * ((lane * 3 + i) % 11)); }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * lane + i;
    const float kf = 0.999f;
    // a weight-ring-like load stream next to the stores: one 16-byte-per-lane load per iteration out of 2.6 MB (L2), consumed DIST
    // iterations later -- the s_waitcnt vmcnt(n) in front of the consumer also waits for every store issued BEFORE that load
    f32x4 ring[DIST > 0 ? DIST : 1];
    float eat = 0.f;
    if (DIST > 0) {
#pragma unroll
        for (int d = 0; d < DIST; ++d) ring[d] = stream[(size_t)((blockIdx.x * 8 + d) % 2560) * 64 + lane];
    }
    for (int it0 = 0; it0 < iters; it0 += (DIST > 0 ? DIST : 1))
#pragma unroll
    for (int dd = 0; dd < (DIST > 0 ? DIST : 1); ++dd) {
        const int it = it0 + dd;
        if (DIST > 0) {
            eat += ring[dd][0];
            ring[dd] = stream[(size_t)((it * 7 + blockIdx.x) % 2560) * 64 + lane];
        }
        float* p = base + (size_t)(RESIDENT ? (it & 7) : it) * (MODE == 5 ? 128 : 256);      // 16 bytes per lane and iteration: 1 KiB per wave (RESIDENT: 8 KiB per wave, stays in L2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(6 * t + i) % 8]) : "v"(kf));
            if (MODE == 1) __builtin_nontemporal_store(v[t], &p[t * 64 + lane]);
            if (MODE == 5 && (t & 1) == 0) __builtin_nontemporal_store(v[t], &p[(t >> 1) * 64 + lane]);     // half the store density
            if (MODE == 2 && (t & 1)) { f32x2 x = {v[t - 1], v[t]}; __builtin_nontemporal_store(x, (f32x2*)&p[(t >> 1) * 128 + 2 * lane]); }
            if (MODE == 3 && t == 3) { f32x4 x = {v[0], v[1], v[2], v[3]}; __builtin_nontemporal_store(x, (f32x4*)&p[4 * lane]); }
            if (MODE == 4) {
                const uint32_t off = (uint32_t)((t * 64 + lane) * 4);
                asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(off), "v"(v[t]), "s"(p) : "memory");
            }
        }
    }
    float s = eat;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += acc[t][0];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, bool RESIDENT = false, int DIST = 0>
static void run(const char* what, float* buf, size_t floats_per_wave, int iters, float* sink, int grid) {
    static f32x4* stream = nullptr;
    if (!stream) { (void)hipMalloc(&stream, 2560 * 1024); (void)hipMemset(stream, 0, 2560 * 1024); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto k = busy<MODE, RESIDENT, DIST>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, floats_per_wave, iters / 8, sink, stream);
    (void)hipMemset(buf, 0xff, (size_t)grid * 4 * floats_per_wave * 4);      // NaN pattern: every float a store should replace
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, floats_per_wave, iters, sink, stream);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (MODE && MODE != 5 && !RESIDENT) {                       // spot check: the first and the last wave's first and last 4 KiB
        size_t bad = 0;
        static float h[1024];
        for (size_t w : {(size_t)0, (size_t)grid * 4 - 1})
            for (size_t it : {(size_t)0, (size_t)iters - 1}) {
                (void)hipMemcpy(h, buf + w * floats_per_wave + it * 256, 1024, hipMemcpyDeviceToHost);
                size_t b0 = 0;
                for (int i = 0; i < 256; ++i) b0 += (h[i] != h[i]);
                if (b0) printf("   wave %zu iteration %zu: %zu of 256 floats not written (first missing index %d)\n", w, it, b0, [&] { for (int i = 0; i < 256; ++i) if (h[i] != h[i]) return i; return -1; }());
                bad += b0;
            }
        if (bad) printf("   !! %zu of 1024 checked floats were not written\n", bad);
    }
    const double bytes = MODE ? (double)grid * 4 * iters * (MODE == 5 ? 512 : 1024) : 0;
    printf("%-44s %7.3f ms  %6.1f ns per iteration (4 MFMAs)  %5.2f TB/s stored  (%s)\n", what, ms, ms * 1e6 / iters, bytes / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = 2 * cus, iters = 6000;                     // 2048 waves x 6000 x 1 KiB = 12.6 GB
    const size_t floats_per_wave = (size_t)iters * 256;
    float *buf, *sink;
    if (hipMalloc(&buf, (size_t)grid * 4 * floats_per_wave * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&sink, (size_t)grid * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("no stores", buf, floats_per_wave, iters, sink, grid);
        run<1>("4 x global_store_dword (as the kernels)", buf, floats_per_wave, iters, sink, grid);
        run<2>("2 x global_store_dwordx2", buf, floats_per_wave, iters, sink, grid);
        run<3>("1 x global_store_dwordx4", buf, floats_per_wave, iters, sink, grid);
        run<4>("4 x global_store_dword, saddr form", buf, floats_per_wave, iters, sink, grid);
        run<0, false, 2>("no stores, ring load consumed 2 iterations later", buf, floats_per_wave, iters, sink, grid);
        run<1, false, 2>("4 x dword + ring load consumed 2 later", buf, floats_per_wave, iters, sink, grid);
        run<1, false, 4>("4 x dword + ring load consumed 4 later", buf, floats_per_wave, iters, sink, grid);
        run<1, false, 8>("4 x dword + ring load consumed 8 later", buf, floats_per_wave, iters, sink, grid);
        run<1, false, 12>("4 x dword + ring load consumed 12 later", buf, floats_per_wave, iters, sink, grid);
        // the kernels' regime: store time about equal to compute time (half the density above)
        run<5>("2 x dword per iteration, no ring", buf, floats_per_wave, iters, sink, grid);
        run<5, false, 2>("2 x dword + ring load consumed 2 later", buf, floats_per_wave, iters, sink, grid);
        run<5, false, 8>("2 x dword + ring load consumed 8 later", buf, floats_per_wave, iters, sink, grid);
        run<1, true>("L2-resident: 4 x global_store_dword", buf, floats_per_wave, iters, sink, grid);
        run<2, true>("L2-resident: 2 x global_store_dwordx2", buf, floats_per_wave, iters, sink, grid);
        run<3, true>("L2-resident: 1 x global_store_dwordx4", buf, floats_per_wave, iters, sink, grid);
        run<4, true>("L2-resident: 4 x dword, saddr form", buf, floats_per_wave, iters, sink, grid);
    }
    return 0;
}
