// Fused field forward with the trunk layers 2..8 as bf16x6 split products (sigma-only coarse pass, Wd = 256).
// Same function as field_fwd_kernel<256,1,SIGMA,FREQ10> (script/models/nerfh_nff.py:192-202,525-555 + rendering.py:114):
// pts = o + d*z -> frequency embedding -> 8-layer skip MLP -> heads.  Every product runs on v_mfma_f32_32x32x16_bf16:
//     x = xh + xm + xl, w = wh + wm + wl exactly (three bf16 each: 24 = 3 x 8 mantissa bits, truncation split)
//     w x ~= wh xh + wh xm + wm xh + wh xl + wl xh + wm xm        (dropped terms: relative size 2^-24)
// accumulated in fp32 by the matrix core: fp32-level accuracy (tools/bf16x6_accuracy.py: 2.5e-7 of the output scale
// against 4.7e-7 for a plain fp32 GEMM) at 6 x 32 = 192 cycles per 16 k-values and tile instead of 8 x 64 = 512.
// The accumulator -> B-operand chaining carries over: a 32-row D tile is two K=16 steps (registers 8j..8j+7 of lane
// group g hold rows rho_g(8j+i)); the host packer emits the weight triples in that order (pack.cpp, x6 segments).
// The bf16 MFMA holds the vector issue port for a quarter of its time, so the split (5.5 VALU per element) hides in the
// gaps (tools/probe/overlap_probe.hip).  Weight stream: 48 KiB slabs = 16 units of three 1 KiB groups (hi, mid, lo).
#define NEFES_SLAB_KIB NEFES_X6_SLAB_KIB
// every run of compiler-placed MFMAs ends with field_common.h mfma_results_fence: hipcc pads an MFMA's result against its own vector
// instructions, but takes the first path it finds to the MFMA where two join (seen: 7 of 18 wait states on the path that skips the
// mask stores) and does not look into the asm statements of the operand functors at all (tools/hazard_lint.py rules B1 / B2)
#define NEFES_ASM_READS_ACC
#include "field_common.h"
#include "../../include/nefes_hip.h"

#include "field_x6.h"
#define NEFES_X6_SLOTS 3   // 144 KiB ring

struct FieldFwdX6Args {
    const char* stream;
    const float* bias;
    uint32_t n_slabs, bias_floats;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    const float* xyz_enc;    // [M,32] (NEFES_XYZ_EXTERNAL32) or null
    const float* viewdirs;   // [N,3] (FULL)
    float* raw_t;            // [N][R][S]
    uint32_t* masks;         // [tiles32][MW][64] or null (FULL)
    int N, S, R, C;
    long long M;
    int n_tiles;
};

// MODE: NEFES_FIELD_SIGMA or NEFES_FIELD_FULL; ENC: NEFES_XYZ_FREQ10 or NEFES_XYZ_EXTERNAL32 (hash grid);
// (W, NTR) = (256, 1) [C = 16] or (128, 5) [C = 128: the reference-default shape]
template <int MODE, int ENC, int W = 256, int NTR = 1, int NP = 6>
__global__ __launch_bounds__(256, 1) void field_fwd_x6_kernel(FieldFwdX6Args a) {
    constexpr int NTW = W / 32, NTH = W / 64, HS = W / 2, GS = W / 4;
    constexpr int ES = ENC == NEFES_XYZ_EXTERNAL32 ? NEFES_X_STEPS : NEFES_E_STEPS;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128), WT = (NTW + 1) / 2, WH = (NTH + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring_base = smem;
    float* bias_lds = (float*)(smem + NEFES_X6_SLOTS * NEFES_SLAB_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    for (uint32_t i = threadIdx.x; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
    WeightRing<NEFES_X6_SLOTS> ring;
    ring.init(a.stream, a.n_slabs, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_base, wave, lane);
    __syncthreads();
    const char* ring_lane = ring_base + lane * 16;
    const char* bias_half = (const char*)bias_lds + 16 * h;
    ring.prime(ring_lane);
    // bias block offsets (floats), in stream order: L1..L8, SIG, FINAL, DIR, RGB, T0, T1, T2, TH (as in field_fwd.hip)
    constexpr int B_SIG = 8 * W, B_FINAL = B_SIG + 32, B_DIR = B_FINAL + W, B_RGB = B_DIR + W / 2,
                  B_T0 = B_RGB + 32 * NTR, B_T1 = B_T0 + W / 2, B_T2 = B_T1 + W / 2, B_TH = B_T2 + W / 2;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long m_raw = (long long)tile * 128 + wave * 32 + j;
        const bool valid = m_raw < a.M;
        const long long m = valid ? m_raw : a.M - 1;
        const int ray = (int)(m / a.S);
        const int smp = (int)(m - (long long)ray * a.S);
        float in_o[3] = {0.f, 0.f, 0.f}, in_d[3] = {0.f, 0.f, 0.f}, in_z = 0.f, v[3] = {0.f, 0.f, 0.f};
        float E[ES];
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
#pragma unroll
            for (int s = 0; s < ES; ++s) E[s] = a.xyz_enc[m * 32 + 2 * s + h];    // compact slots: feature 2s+h
        } else if (a.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) in_o[c] = a.pts[m * 3 + c];
        } else {
            in_z = a.z[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) { in_o[c] = a.rays_o[ray * 3 + c]; in_d[c] = a.rays_d[ray * 3 + c]; }
        }
        if (MODE != NEFES_FIELD_SIGMA) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[ray * 3 + c];
        }
        loads_landed();
        pin(in_o); pin(in_d); pin(in_z); pin(v);
        if constexpr (ENC == NEFES_XYZ_EXTERNAL32) {
            pin(E);
        } else {
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = a.pts ? in_o[c] : add_rn(in_o[c], mul_rn(in_d[c], in_z));   // rendering.py:114,142
            embed_slots<NEFES_N_FREQ_XYZ>(E, x, h);
        }
        uint32_t* mask_tile = (MODE == NEFES_FIELD_FULL && a.masks)
                                  ? a.masks + ((size_t)((long long)tile * 4 + wave) * MW) * 64
                                  : nullptr;
        int mask_word = 0;
        auto put_masks = [&](const uint32_t* bits, int n) {
            if (mask_tile) {
                for (int w = 0; w < n; ++w) mask_tile[(mask_word + w) * 64 + lane] = bits[w];
                mask_word += n;
            }
        };
        const size_t raw_off = (size_t)ray * a.R * a.S + smp;
        auto raw_col = [&]() {
            float* pcol = a.raw_t + raw_off;
            asm volatile("" : "+v"(pcol));
            return pcol;
        };
        auto bias_at = [&](int off_floats) { return BiasInit{bias_half + off_floats * 4}; };
        const ArrayIn<ES> in_E{E};
        f32x16 A[NTW], B[NTW];
        uint32_t bits[WT];
        auto clear_bits = [&]() {
#pragma unroll
            for (int w = 0; w < WT; ++w) bits[w] = 0u;
        };
        constexpr bool CAP = MODE == NEFES_FIELD_FULL;
        auto sigma_head = [&](const f32x16 (&X)[NTW]) {
            f32x16 sg[1];
            mma_run_x6<1, W / 16, 0, true, NP>(ring, ring_lane, ReluSplit<false, NTW, WT>{X, bits}, bias_at(B_SIG), sg);   // static_sigma
            if (valid && h == 0) {
                const int ch = (MODE == NEFES_FIELD_SIGMA) ? 0 : 3 + a.C;
                raw_col()[(size_t)ch * a.S] = softplus_ref(sg[0][0]);
            }
        };
        mma_run_x6<NTW, ES / 8, 0, true, NP>(ring, ring_lane, ArraySplit<ES>{E}, bias_at(0), A);               // layer 1
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            clear_bits();
            mma_run_x6<NTW, W / 16, 0, true, NP>(ring, ring_lane, ReluSplit<CAP, NTW, WT>{A, bits}, bias_at((l1 - 1) * W), B);   // layers 2, 4, 6, 8
            put_masks(bits, WT);                                                                      // mask of layer l1-1
            if (p == 3) {
                if (MODE == NEFES_FIELD_SIGMA) break;
                sigma_head(B);
            }
            clear_bits();
            mma_run_x6<NTW, W / 16, 0, true, NP>(ring, ring_lane, ReluSplit<CAP, NTW, WT>{B, bits}, bias_at(l2 <= 8 ? (l2 - 1) * W : B_FINAL), A);   // 3, 5, 7, final
            if (p == 1) mma_run_x6<NTW, ES / 8, 0, false, NP>(ring, ring_lane, ArraySplit<ES>{E}, ZeroInit{}, A);   // skip: + W5[:, :63] e
            put_masks(bits, WT);                                                                      // mask of layer l1
        }
        if constexpr (MODE == NEFES_FIELD_SIGMA) sigma_head(B);
        if constexpr (MODE == NEFES_FIELD_FULL) {
            // dir_encoding and transient_encoding.0 as ONE stacked 2*NTH-tile product (pack.cpp add_heads_x6): tiles
            // [0, NTH) = dir, [NTH, 2 NTH) = t0
            float Dv[16];                                              // 14 embedding slots + 2 padding slots (two 16-k steps)
            {
                float d14[NEFES_D_STEPS];
                embed_slots<NEFES_N_FREQ_DIR>(d14, v, h);
#pragma unroll
                for (int s = 0; s < 16; ++s) Dv[s] = s < NEFES_D_STEPS ? d14[s] : 0.f;
            }
            f32x16 dt[2 * NTH], acc3[NTH], acc2[NTH];
            uint32_t bits2[WH];
            auto clear2 = [&]() {
#pragma unroll
                for (int w = 0; w < WH; ++w) bits2[w] = 0u;
            };
            struct Bias2 {             // C operands of the stacked product: dir bias tiles, then t0 bias tiles
                BiasInit a, b;
                __device__ __forceinline__ f32x16 operator()(int t) const { return t < NTH ? a(t) : b(t - NTH); }
            };
            mma_run_x6<2 * NTH, W / 16, 0, true, NP>(ring, ring_lane, IdentSplit<NTW, 0>{A}, Bias2{bias_at(B_DIR), bias_at(B_T0)}, dt);
            mma_run_x6<2 * NTH, 2, 0, false, NP>(ring, ring_lane, ArraySplit<16>{Dv}, ZeroInit{}, dt);
            {
                f32x16 ar[NTR];
                clear2();
                mma_run_x6<NTR, W / 32, 0, true, NP>(ring, ring_lane, ReluSplit<true, 2 * NTH, WH>{dt, bits2}, bias_at(B_RGB), ar);
                put_masks(bits2, WH);                                 // dir_encoding
                if (valid) {
                    float* ph = raw_col() + (size_t)(4 * h) * a.S;
#pragma unroll
                    for (int t = 0; t < NTR; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int cu = 32 * t + nefes_rho(0, r);
                            if (cu + 4 * h < 3 + a.C) ph[(size_t)cu * a.S] = ar[t][r];
                        }
                }
            }
            clear2();
            mma_run_x6<NTH, W / 32, 0, true, NP>(ring, ring_lane, ReluSplit<true, 2 * NTH, WH, NTH>{dt, bits2}, bias_at(B_T1), acc3);
            put_masks(bits2, WH);                                     // transient_encoding.0
            clear2();
            mma_run_x6<NTH, W / 32, 0, true, NP>(ring, ring_lane, ReluSplit<true, NTH, WH>{acc3, bits2}, bias_at(B_T2), acc2);
            put_masks(bits2, WH);                                     // transient_encoding.2
            f32x16 th[1];
            clear2();
            mma_run_x6<1, W / 32, 0, true, NP>(ring, ring_lane, ReluSplit<true, NTH, WH>{acc2, bits2}, bias_at(B_TH), th);
            put_masks(bits2, WH);
            if (valid) {
                float* o = raw_col() + (size_t)(3 + a.C + 1) * a.S;
                if (h == 0) {
                    o[0] = sigmoid_ref(th[0][0]);
                    o[(size_t)a.S] = sigmoid_ref(th[0][1]);
                    o[(size_t)2 * a.S] = sigmoid_ref(th[0][2]);
                    o[(size_t)3 * a.S] = softplus_ref(th[0][3]);
                } else {
                    o[(size_t)4 * a.S] = softplus_ref(th[0][0]);
                }
            }
        }
    }
    ring.drain();
}

template <int MODE, int ENC, int W = 256, int NTR = 1, int NP = 6>
static int launch_x6(const FieldFwdX6Args& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_X6_SLOTS * NEFES_SLAB_BYTES + ((a.bias_floats * 4 + 255) / 256) * 256;
    auto k = field_fwd_x6_kernel<MODE, ENC, W, NTR, NP>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = a.n_tiles < cus ? a.n_tiles : cus;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

// Kernel instances spread over two objects built from this one source (Makefile: -DNEFES_TU_PART=0..1): part 0 = entry points
// + the Wd = 256 frequency-embedding instances, part 1 = hash-grid and Wd = 128 instances.
#ifndef NEFES_TU_PART
#define NEFES_TU_PART 0
#endif
enum { FWD_EXT_SIGMA = 0, FWD_EXT_FULL, FWD_128_SIGMA, FWD_128_FULL };
int nefes_fwd_x6_launch_part1(int which, const FieldFwdX6Args& a, hipStream_t st);

#if NEFES_TU_PART == 1
int nefes_fwd_x6_launch_part1(int which, const FieldFwdX6Args& a, hipStream_t st) {
    switch (which) {
        case FWD_EXT_SIGMA: return launch_x6<NEFES_FIELD_SIGMA, NEFES_XYZ_EXTERNAL32>(a, st);
        case FWD_EXT_FULL: return launch_x6<NEFES_FIELD_FULL, NEFES_XYZ_EXTERNAL32>(a, st);
        case FWD_128_SIGMA: return launch_x6<NEFES_FIELD_SIGMA, NEFES_XYZ_FREQ10, 128, 5>(a, st);
        case FWD_128_FULL: return launch_x6<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10, 128, 5>(a, st);
    }
    return NEFES_E_UNSUPPORTED;
}
#else   // part 0

static int field_fwd_x6_impl(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                             const float* rays_d, const float* z, const float* pts, const float* xyz_enc,
                             const float* viewdirs, float* raw_t, uint32_t* masks, void* stream) {
    if (!desc || !packed || !raw_t || N <= 0 || S <= 0) return NEFES_E_BADARG;
    const bool ext = desc->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    if (ext ? !xyz_enc : (!pts && !(rays_o && rays_d && z))) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_SIGMA && mode != NEFES_FIELD_FULL) return NEFES_E_UNSUPPORTED;
    if (mode == NEFES_FIELD_FULL && (!viewdirs || !desc->has_transient)) return NEFES_E_BADARG;
    const bool big = desc->width == 256 && desc->feat_dim == 16, small = desc->width == 128 && desc->feat_dim == 128 && !ext;
    if (!(big || small) || (desc->xyz_encoding != NEFES_XYZ_FREQ10 && !ext)) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[mode == NEFES_FIELD_SIGMA ? NEFES_STREAM_FWD_SIGMA_X6 : NEFES_STREAM_FWD_FULL_X6];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldFwdX6Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.bias = (const float*)((const char*)packed + si.bias_off);
    a.n_slabs = si.n_slabs; a.bias_floats = si.bias_floats;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.xyz_enc = xyz_enc; a.viewdirs = viewdirs; a.raw_t = raw_t; a.masks = masks;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = mode == NEFES_FIELD_SIGMA ? 1 : 3 + a.C + 6;
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    if (small) {
        if (mode == NEFES_FIELD_SIGMA) return nefes_fwd_x6_launch_part1(FWD_128_SIGMA, a, st);
        return nefes_fwd_x6_launch_part1(FWD_128_FULL, a, st);
    }
    if (ext) {
        if (mode == NEFES_FIELD_SIGMA) return nefes_fwd_x6_launch_part1(FWD_EXT_SIGMA, a, st);
        return nefes_fwd_x6_launch_part1(FWD_EXT_FULL, a, st);
    }
    if (mode == NEFES_FIELD_SIGMA) return launch_x6<NEFES_FIELD_SIGMA, NEFES_XYZ_FREQ10>(a, st);
    return launch_x6<NEFES_FIELD_FULL, NEFES_XYZ_FREQ10>(a, st);
}

extern "C" int nefes_field_fwd_x6(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* xyz_enc,
                                  const float* viewdirs, float* raw_t, uint32_t* masks, void* stream) {
    return field_fwd_x6_impl(desc, packed, mode, N, S, rays_o, rays_d, z, pts, xyz_enc, viewdirs, raw_t, masks, stream);
}
#endif   // NEFES_TU_PART
