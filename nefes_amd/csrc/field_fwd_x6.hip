// Fused field forward with the trunk layers 2..8 as bf16x6 split products (sigma-only coarse pass, Wd = 256).
// Same function as field_fwd_kernel<256,1,SIGMA,FREQ10> (script/models/nerfh_nff.py:192-202,525-555 + rendering.py:114):
// pts = o + d*z -> frequency embedding -> 8-layer skip MLP -> static_sigma.  Layer 1, the skip's embedding part and the
// sigma head stay on v_mfma_f32_32x32x2_f32; the seven 256x256 hidden products run on v_mfma_f32_32x32x16_bf16:
//     x = xh + xm + xl, w = wh + wm + wl exactly (three bf16 each: 24 = 3 x 8 mantissa bits, truncation split)
//     w x ~= wh xh + wh xm + wm xh + wh xl + wl xh + wm xm        (dropped terms: relative size 2^-24)
// accumulated in fp32 by the matrix core: fp32-level accuracy (tools/bf16x6_accuracy.py: 2.5e-7 of the output scale
// against 4.7e-7 for a plain fp32 GEMM) at 6 x 32 = 192 cycles per 16 k-values and tile instead of 8 x 64 = 512.
// The accumulator -> B-operand chaining carries over: a 32-row D tile is two K=16 steps (registers 8j..8j+7 of lane
// group g hold rows rho_g(8j+i)); the host packer emits the weight triples in that order (pack.cpp, x6 segments).
// The bf16 MFMA holds the vector issue port for a quarter of its time, so the split (5.5 VALU per element) hides in the
// gaps (tools/probe/overlap_probe.hip).  Weight stream: 48 KiB slabs = 16 units of three 1 KiB groups (hi, mid, lo).
#define NEFES_SLAB_KIB NEFES_X6_SLAB_KIB
#include "field_common.h"
#include "../../include/nefes_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define NEFES_X6_SLOTS 3   // 144 KiB ring

struct FieldFwdX6Args {
    const char* stream;
    const float* bias;
    uint32_t n_slabs, bias_floats;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    const float* viewdirs;   // [N,3] (FULL)
    float* raw_t;            // [N][R][S]
    uint32_t* masks;         // [tiles32][MW][64] or null (FULL)
    int N, S, R, C;
    long long M;
    int n_tiles;
};

struct Split3 {
    u32x4 h, m, l;           // 8 bf16 each: element i in the low/high half of dword i/2
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// relu(X[8j .. 8j+7] of tile T) -> three packed bf16 vectors (hi, mid, lo), x = hi + mid + lo exactly.
// CAPTURE: also shift the sign bit of each pre-activation into the layer's ReLU-mask words, in the same order as the fp32
// kernel (activation 8*s16 + i <-> k-step 16T + r there), so field_bwd_kernel reads identical words.
template <bool CAPTURE, int NX, int NWORDS>
__device__ __forceinline__ Split3 split_relu(const f32x16 (&X)[NX], int s16, uint32_t (&bits)[NWORDS]) {
    const int T = s16 >> 1, r0 = (s16 & 1) * 8;
    Split3 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float v0 = X[T][r0 + 2 * p], v1 = X[T][r0 + 2 * p + 1];
        if (CAPTURE) {
            mask_shift_in(bits[(8 * s16 + 2 * p) >> 5], v0);
            mask_shift_in(bits[(8 * s16 + 2 * p + 1) >> 5], v1);
        }
        const float x0 = fmaxf(v0, 0.f), x1 = fmaxf(v1, 0.f);
        const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
        const float e0 = x0 - __uint_as_float(b0 & 0xffff0000u), e1 = x1 - __uint_as_float(b1 & 0xffff0000u);   // exact
        const uint32_t c0 = __float_as_uint(e0), c1 = __float_as_uint(e1);
        const float f0 = e0 - __uint_as_float(c0 & 0xffff0000u), f1 = e1 - __uint_as_float(c1 & 0xffff0000u);   // exact, <= 8 bits
        // v_perm_b32: bytes {src0 = element 1, src1 = element 0}; take the upper halves -> (hi16(x1) << 16) | hi16(x0)
        o.h[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        o.m[p] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
        o.l[p] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
    }
    return o;
}

// acc[0..NT) = W * relu(X) over KS16 steps of 16 k-values, bias as the C operand of each tile's first MFMA.
// Stream order: for k16-step q, for tile t: [A_hi | A_mid | A_lo] (3 KiB unit); 16 units per 48 KiB slab.
template <int NT, int KS16, bool CAPTURE, class InitFn, int NX, int NWORDS, int SLOTS>
__device__ __forceinline__ void mma_run_x6(WeightRing<SLOTS>& ring, const char* ring_lane, const f32x16 (&X)[NX],
                                           uint32_t (&bits)[NWORDS], const InitFn& init, f32x16 (&acc)[NT]) {
    constexpr int UPS = (NEFES_SLAB_FRAGS / 4) / 3;       // units per slab
    constexpr int NU = KS16 * NT;
    constexpr int NSLAB = (NU + UPS - 1) / UPS;
    Split3 B = split_relu<CAPTURE>(X, 0, bits), Bn = B;
    f32x16 c0 = init(0);                                   // bias tile of the next first-step unit, fetched one unit ahead
    const char* p = ring_lane + ring.cur_off;
    f32x4 ah = ring.pf, am = *(const f32x4*)(p + 1024), al = *(const f32x4*)(p + 2048);
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int nu = (NU - sl * UPS) < UPS ? (NU - sl * UPS) : UPS;
#pragma unroll
        for (int uu = 0; uu < UPS; ++uu) {
            if (uu < nu) {
                const int u = sl * UPS + uu, q = u / NT, t = u % NT;
                // A operands of the next unit (of this slab, or of the slab acquired here: the stream is one sequence)
                f32x4 nh, nm, nl;
                if (uu + 1 < nu) {
                    nh = *(const f32x4*)(p + (3 * uu + 3) * 1024);
                    nm = *(const f32x4*)(p + (3 * uu + 4) * 1024);
                    nl = *(const f32x4*)(p + (3 * uu + 5) * 1024);
                } else {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES >= uu) ring.issue_piece(qq);   // everything still owed to this slab
                    ring.cur_off = ring.acquire();
                    p = ring_lane + ring.cur_off;
                    nh = *(const f32x4*)(p);
                    nm = *(const f32x4*)(p + 1024);
                    nl = *(const f32x4*)(p + 2048);
                }
                if (t == 0 && q > 0) B = Bn;
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 Ah = as_bf16x8(ah), Am = as_bf16x8(am), Al = as_bf16x8(al);
                const bf16x8 Bh = as_bf16x8(B.h), Bm = as_bf16x8(B.m), Bl = as_bf16x8(B.l);
                f32x16 c;
                if (q == 0) {
                    c = c0;
                    if (t + 1 < NT) c0 = init(t + 1);
                } else {
                    c = acc[t];
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, c, 0, 0, 0);      // small terms first
                if (uu + 1 < nu) {
#pragma unroll
                    for (int qq = 0; qq < NEFES_SLAB_PIECES; ++qq)
                        if ((qq * nu) / NEFES_SLAB_PIECES == uu) {
                            __builtin_amdgcn_sched_barrier(0);
                            ring.issue_piece(qq);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, c, 0, 0, 0);
                acc[t] = c;
                // the next k16-step's operand: split in the gaps of this step's MFMAs (one eighth per tile would be finer;
                // one block in the middle of the step is what the probe measured as hidden)
                if (t == NT / 2 && q + 1 < KS16) Bn = split_relu<CAPTURE>(X, q + 1, bits);
                ah = nh; am = nm; al = nl;
            }
        }
    }
    ring.pf = ah;
}

template <int MODE>   // NEFES_FIELD_SIGMA or NEFES_FIELD_FULL; Wd = 256, C = 16, frequency embedding
__global__ __launch_bounds__(256, 1) void field_fwd_x6_kernel(FieldFwdX6Args a) {
    constexpr int W = 256, NTW = 8, NTH = 4, NTR = 1, HS = W / 2, GS = W / 4, ES = NEFES_E_STEPS;
    constexpr int MW = 8 * (W / 64) + 4 * (W / 128), WT = 4, WH = 2;
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
        if (a.pts) {
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
        float E[ES], x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = a.pts ? in_o[c] : add_rn(in_o[c], mul_rn(in_d[c], in_z));   // rendering.py:114,142
        embed_slots<NEFES_N_FREQ_XYZ>(E, x, h);
        float Dv[NEFES_D_STEPS];
        if (MODE != NEFES_FIELD_SIGMA) embed_slots<NEFES_N_FREQ_DIR>(Dv, v, h);
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
        const ArrayIn<NEFES_D_STEPS> in_D{Dv};
        f32x16 A[NTW], B[NTW];
        uint32_t bits[WT];
        auto clear_bits = [&]() {
#pragma unroll
            for (int w = 0; w < WT; ++w) bits[w] = 0u;
        };
        constexpr bool CAP = MODE == NEFES_FIELD_FULL;
        auto sigma_head = [&](const f32x16 (&X)[NTW]) {
            f32x16 sg[1];
            mma_run<1, HS, 0, true>(ring, ring_lane, ReluIn<NTW>{X}, bias_at(B_SIG), sg);             // static_sigma (fp32)
            if (valid && h == 0) {
                const int ch = (MODE == NEFES_FIELD_SIGMA) ? 0 : 3 + a.C;
                raw_col()[(size_t)ch * a.S] = softplus_ref(sg[0][0]);
            }
        };
        mma_run<NTW, ES, 0, true>(ring, ring_lane, in_E, bias_at(0), A);                              // layer 1 (fp32)
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int l1 = 2 + 2 * p, l2 = l1 + 1;
            clear_bits();
            mma_run_x6<NTW, W / 16, CAP>(ring, ring_lane, A, bits, bias_at((l1 - 1) * W), B);         // layers 2, 4, 6, 8
            put_masks(bits, WT);                                                                      // mask of layer l1-1
            if (p == 3) {
                if (MODE == NEFES_FIELD_SIGMA) break;
                sigma_head(B);
            }
            clear_bits();
            mma_run_x6<NTW, W / 16, CAP>(ring, ring_lane, B, bits, bias_at(l2 <= 8 ? (l2 - 1) * W : B_FINAL), A);   // 3, 5, 7, final
            if (p == 1) mma_run<NTW, ES, 0, false>(ring, ring_lane, in_E, ZeroInit{}, A);            // skip: + W5[:, :63] e
            put_masks(bits, WT);                                                                      // mask of layer l1
        }
        if constexpr (MODE == NEFES_FIELD_SIGMA) sigma_head(B);
        if constexpr (MODE == NEFES_FIELD_FULL) {
            // the heads run on the fp32 path exactly as in field_fwd_kernel (21 % of the MACs)
            f32x16 acc2[NTH], acc3[NTH];
            uint32_t bits2[WH];
            auto clear2 = [&]() {
#pragma unroll
                for (int w = 0; w < WH; ++w) bits2[w] = 0u;
            };
            mma_run<NTH, HS, 0, true>(ring, ring_lane, IdentIn<NTW>{A}, bias_at(B_DIR), acc2);
            mma_run<NTH, NEFES_D_STEPS, 0, false>(ring, ring_lane, in_D, ZeroInit{}, acc2);
            {
                f32x16 ar[NTR];
                clear2();
                mma_run<NTR, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_RGB), ar);
                put_masks(bits2, WH);
                if (valid) {
                    float* ph = raw_col() + (size_t)(4 * h) * a.S;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cu = nefes_rho(0, r);
                        if (cu + 4 * h < 3 + a.C) ph[(size_t)cu * a.S] = ar[0][r];
                    }
                }
            }
            mma_run<NTH, HS, 0, true>(ring, ring_lane, IdentIn<NTW>{A}, bias_at(B_T0), acc2);
            mma_run<NTH, NEFES_D_STEPS, 0, false>(ring, ring_lane, in_D, ZeroInit{}, acc2);
            clear2();
            mma_run<NTH, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_T1), acc3);
            put_masks(bits2, WH);
            clear2();
            mma_run<NTH, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc3, bits2}, bias_at(B_T2), acc2);
            put_masks(bits2, WH);
            f32x16 th[1];
            clear2();
            mma_run<1, GS, 0, true>(ring, ring_lane, ReluCapture<NTH, WH>{acc2, bits2}, bias_at(B_TH), th);
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

template <int MODE>
static int launch_x6(const FieldFwdX6Args& a, hipStream_t st) {
    const size_t lds = (size_t)NEFES_X6_SLOTS * NEFES_SLAB_BYTES + ((a.bias_floats * 4 + 255) / 256) * 256;
    auto k = field_fwd_x6_kernel<MODE>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = a.n_tiles < cus ? a.n_tiles : cus;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

extern "C" int nefes_field_fwd_x6(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                                  const float* rays_d, const float* z, const float* pts, const float* viewdirs, float* raw_t,
                                  uint32_t* masks, void* stream) {
    if (!desc || !packed || !raw_t || N <= 0 || S <= 0) return NEFES_E_BADARG;
    if (!pts && !(rays_o && rays_d && z)) return NEFES_E_BADARG;
    if (mode != NEFES_FIELD_SIGMA && mode != NEFES_FIELD_FULL) return NEFES_E_UNSUPPORTED;
    if (mode == NEFES_FIELD_FULL && (!viewdirs || !desc->has_transient)) return NEFES_E_BADARG;
    if (desc->width != 256 || desc->feat_dim != 16 || desc->xyz_encoding != NEFES_XYZ_FREQ10) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    int rc = nefes_blob_info(desc, &info);
    if (rc) return rc;
    const NefesStreamInfo& si = info.stream[mode == NEFES_FIELD_SIGMA ? NEFES_STREAM_FWD_SIGMA_X6 : NEFES_STREAM_FWD_FULL_X6];
    if (si.n_slabs == 0) return NEFES_E_UNSUPPORTED;
    FieldFwdX6Args a;
    a.stream = (const char*)packed + si.slab_off;
    a.bias = (const float*)((const char*)packed + si.bias_off);
    a.n_slabs = si.n_slabs; a.bias_floats = si.bias_floats;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.pts = pts; a.viewdirs = viewdirs; a.raw_t = raw_t; a.masks = masks;
    a.N = N; a.S = S; a.C = desc->feat_dim; a.R = mode == NEFES_FIELD_SIGMA ? 1 : 3 + a.C + 6;
    a.M = (long long)N * S;
    a.n_tiles = (int)((a.M + 127) / 128);
    if (mode == NEFES_FIELD_SIGMA) return launch_x6<NEFES_FIELD_SIGMA>(a, (hipStream_t)stream);
    return launch_x6<NEFES_FIELD_FULL>(a, (hipStream_t)stream);
}
