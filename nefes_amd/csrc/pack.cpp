// Host-side weight packer: NeRFH_NFF state_dict (torch [out,in] fp32) -> MFMA fragment streams.
// See layout.h for the vocabulary.  Reference tensors: script/models/nerfh_nff.py:452-505.
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/nefes_hip.h"
#include "layout.h"

namespace {

enum { L_XYZ1 = 0, L_FINAL = 8, L_DIR = 9, L_SIGMA = 10, L_RGB = 11, L_T0 = 12, L_T1 = 13, L_T2 = 14, L_TSIGMA = 15,
       L_TRGB = 16, L_TBETA = 17 };

struct Seg {
    int nt = 0, ks = 0;
    std::vector<int> kidx;  // [ks][2]  column of W (row if transposed) per slot, -1 = pad
    std::vector<int> ridx;  // [nt][32] row of W (column if transposed) per accumulator row, -1 = pad
    const float* W = nullptr;
    int ld = 0;
    bool transposed = false;
    bool x6 = false;        // bf16x6 segment: fragments are bf16 triples (hi, mid, lo) for v_mfma_f32_32x32x16_bf16
    bool h3 = false;        // fp16 two-part segment: (hi, lo) fp16 pairs of W * 2^wexp for v_mfma_f32_32x32x16_f16 (set with x6)
    int wexp = 0;           // h3: power-of-two exponent the segment's weights are stored with
    float at(int s, int t, int lane) const {
        const int i = lane & 31, h = lane >> 5;
        const int r = ridx[t * 32 + i], k = kidx[s * 2 + h];
        if (r < 0 || k < 0) return 0.f;
        return transposed ? W[(size_t)k * ld + r] : W[(size_t)r * ld + k];
    }
    // x6: element (k16-step q, lane group g, i) of the A operand = k-step 8q+i, half g of the fp32 ordering
    float at16(int q, int i, int t, int lane) const {
        const int m = lane & 31, g = lane >> 5;
        const int r = ridx[t * 32 + m], k = kidx[(8 * q + i) * 2 + g];
        if (r < 0 || k < 0) return 0.f;
        return transposed ? W[(size_t)k * ld + r] : W[(size_t)r * ld + k];
    }
    // max over the segment's output rows of sum_k |w|: |W x|_inf <= row_bound |x|_inf (the fp16 kernels scale the NEXT operand
    // from this bound instead of scanning the accumulators for their maximum)
    float row_bound() const {
        if (!W) return 0.f;
        float best = 0.f;
        for (int t = 0; t < nt; ++t)
            for (int i = 0; i < 32; ++i) {
                const int r = ridx[t * 32 + i];
                if (r < 0) continue;
                double sum = 0.0;
                for (int s = 0; s < ks; ++s)
                    for (int hh = 0; hh < 2; ++hh) {
                        const int kk = kidx[s * 2 + hh];
                        if (kk >= 0) sum += fabs((double)(transposed ? W[(size_t)kk * ld + r] : W[(size_t)r * ld + kk]));
                    }
                if ((float)sum > best) best = (float)sum;
            }
        return best * 1.0001f;
    }
    int units() const { return (ks / 8) * nt; }                       // x6: one unit = (k16-step, tile) = 3 KiB; h3: 2 KiB
    int unit_kib() const { return h3 ? 2 : 3; }
    int slabs(int slab_frags) const {
        if (!x6) return nefes_segment_slabs(nt, ks, slab_frags);
        const int ups = (slab_frags / 4) / unit_kib();
        return (units() + ups - 1) / ups;
    }
};

// w = hi + mid + lo exactly, each a bf16: hi = RNE(w), mid = RNE(w - hi), lo = w - hi - mid (both residuals are exact in
// fp32, the last one has at most 8 significant bits).  Round-to-nearest parts make hi + mid an unbiased 16-bit value of w,
// (an earlier three-product experiment consumed that); the six-product kernels see an exact triple either way.
static uint16_t rne_bf16(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
static void split_bf16x3(float w, uint16_t (&part)[3]) {
    float r = w;
    for (int p = 0; p < 3; ++p) {
        part[p] = rne_bf16(r);
        const uint32_t b = (uint32_t)part[p] << 16;
        float f;
        memcpy(&f, &b, 4);
        r -= f;
    }
}

// fp32 -> fp16 bits, round to nearest even, subnormals kept (what v_cvt_f16_f32 / v_fma_mixlo_f16 do in the kernels)
static uint16_t rne_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));     // >= 2^16: inf / nan
    if (x < 0x38800000u) {                                                                     // < 2^-14: subnormal or zero
        float a;
        memcpy(&a, &x, 4);
        return (uint16_t)(sign | (uint32_t)nearbyintf(a * 16777216.0f));                        // units of 2^-24, ties to even
    }
    const uint32_t mant = x & 0x7fffffu, e = (x >> 23) - 112u;
    uint32_t h = (e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                                     // a carry walks into the exponent
    return (uint16_t)(sign | h);
}
static float f16_value(uint16_t h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(1024 + m), e - 25);
    return (h & 0x8000u) ? -v : v;
}
// w * 2^e = hi + lo (+ a remainder below 2^-22 of |w| 2^e): hi = RNE_f16(w 2^e), lo = RNE_f16(w 2^e - hi); the difference is
// exact in fp32
static void split_f16x2(float w, int e, uint16_t (&part)[2]) {
    const float y = ldexpf(w, e);
    part[0] = rne_f16(y);
    part[1] = rne_f16(y - f16_value(part[0]));
}
// exponent e with max|w| * 2^e in [2^NEFES_H3_TARGET_EXP, 2^(NEFES_H3_TARGET_EXP+1))
static int scale_exp(float amax) {
    if (!(amax > 0.f) || !isfinite(amax)) return 0;
    int e;
    frexpf(amax, &e);                          // amax = m 2^e, m in [0.5, 1)
    int r = NEFES_H3_TARGET_EXP + 1 - e;
    return r < -60 ? -60 : (r > 60 ? 60 : r);
}

struct BiasBlk {
    const float* b;
    std::vector<int> ridx;  // [nt*32] index into b or -1
};

struct Stream {
    std::vector<Seg> segs;
    std::vector<BiasBlk> bias;
    bool h3 = false;        // carries a weight-scale exponent table (one int32 per segment) behind the bias blocks
    int n_slabs(int slab_frags) const {
        int n = 0;
        for (auto& s : segs) n += s.slabs(slab_frags);
        return n;
    }
    int bias_only() const {
        int n = 0;
        for (auto& b : bias) n += (int)b.ridx.size();
        return n;
    }
    int scale_count() const { return h3 ? (2 * (int)segs.size() + (int)bias.size() + 3) / 4 * 4 : 0; }
    int bias_floats() const { return bias_only() + scale_count(); }
};

std::vector<int> rows_natural(int nt, int limit) {
    std::vector<int> r(nt * 32);
    for (int i = 0; i < nt * 32; ++i) r[i] = i < limit ? i : -1;
    return r;
}
std::vector<int> k_natural(int ks, int base) {
    std::vector<int> k(ks * 2);
    for (int s = 0; s < ks; ++s)
        for (int h = 0; h < 2; ++h) k[s * 2 + h] = base + nefes_nat_slot(s, h);
    return k;
}
std::vector<int> k_emb(int L, int ks, int base) {
    std::vector<int> k(ks * 2);
    for (int s = 0; s < ks; ++s)
        for (int h = 0; h < 2; ++h) {
            const int e = nefes_emb_slot(L, s, h);
            k[s * 2 + h] = e < 0 ? -1 : base + e;
        }
    return k;
}
std::vector<int> k_compact(int ks, int limit) {
    std::vector<int> k(ks * 2);
    for (int i = 0; i < ks * 2; ++i) k[i] = i < limit ? i : -1;
    return k;
}
// accumulator rows of embedding tiles (backward): row i of tile tau <-> slot (16*tau + reg(i), half(i))
std::vector<int> rows_emb(int L, int ntiles, int base) {
    std::vector<int> r(ntiles * 32);
    for (int t = 0; t < ntiles; ++t)
        for (int i = 0; i < 32; ++i) {
            const int e = nefes_emb_slot(L, 16 * t + nefes_row_reg(i), nefes_row_half(i));
            r[t * 32 + i] = e < 0 ? -1 : base + e;
        }
    return r;
}
std::vector<int> concat(std::vector<int> a, const std::vector<int>& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}

struct Net {
    int W, W2, C, NTW, NTH, NTR;
    bool transient;
    bool ext;        // xyz embedding supplied by the caller (32 features, e.g. the hash grid) instead of the 63 frequency features
    int in_xyz, e_steps;
    const float* const* t;
    std::vector<float> th_w, th_b;  // virtual transient-head matrix [5][W2]: rgb(3), sigma, beta
    std::vector<float> dt_w;        // virtual stacked [dir_encoding ; transient_encoding.0] matrix [2 W2][W + 27] (x6 streams)
    const float* w(int l) const { return t[2 * l]; }
    const float* b(int l) const { return t[2 * l + 1]; }
};

Seg seg(int nt, int ks, std::vector<int> kidx, std::vector<int> ridx, const float* W, int ld, bool tr = false) {
    Seg s;
    s.nt = nt; s.ks = ks; s.kidx = std::move(kidx); s.ridx = std::move(ridx); s.W = W; s.ld = ld; s.transposed = tr;
    return s;
}

// k-steps / accumulator rows of the xyz embedding: frequency slots (layout.h nefes_emb_slot) or, for an external
// 32-feature embedding, compact slots (feature 2s+h)
std::vector<int> k_xyz(const Net& n, int base) {
    return n.ext ? [&] { auto k = k_compact(n.e_steps, n.in_xyz); for (auto& v : k) if (v >= 0) v += base; return k; }()
                 : k_emb(10, n.e_steps, base);
}
std::vector<int> rows_xyz(const Net& n, int base) {
    if (!n.ext) return rows_emb(10, 2, base);
    // two tiles like the frequency embedding (same kernel shapes): tile 0 = the 32 features, tile 1 = padding
    std::vector<int> r(64, -1);
    for (int i = 0; i < 32; ++i) r[i] = base + 2 * nefes_row_reg(i) + nefes_row_half(i);
    return r;
}

// split kind of a stream's hidden products: 0 = fp32 fragments, 1 = bf16x6 triples, 2 = fp16 (hi, lo) pairs
static void mark(Seg& s, int kind) {
    s.x6 = kind != 0;
    s.h3 = kind == 2;
}

void add_trunk(const Net& n, Stream& st, int x6 = 0) {
    const int W = n.W, NT = n.NTW;
    for (int l = 0; l < 8; ++l) {
        if (l == 0) {
            st.segs.push_back(seg(NT, n.e_steps, k_xyz(n, 0), rows_natural(NT, W), n.w(0), n.in_xyz));
            mark(st.segs.back(), x6);
        } else if (l == 4) {  // skip layer: columns [xyz(63), h(W)]  (nerfh_nff.py:472-473,551-552)
            // the kernels accumulate the hidden part first (its first k-step carries the bias), then the xyz part
            st.segs.push_back(seg(NT, W / 2, k_natural(W / 2, n.in_xyz), rows_natural(NT, W), n.w(4), n.in_xyz + W));
            mark(st.segs.back(), x6);
            st.segs.push_back(seg(NT, n.e_steps, k_xyz(n, 0), rows_natural(NT, W), n.w(4), n.in_xyz + W));
            mark(st.segs.back(), x6);
        } else {
            st.segs.push_back(seg(NT, W / 2, k_natural(W / 2, 0), rows_natural(NT, W), n.w(l), W));
            mark(st.segs.back(), x6);
        }
        st.bias.push_back({n.b(l), rows_natural(NT, W)});
    }
    // static sigma head: one tile, row 0
    st.segs.push_back(seg(1, W / 2, k_natural(W / 2, 0), rows_natural(1, 1), n.w(L_SIGMA), W));
    mark(st.segs.back(), x6);
    st.bias.push_back({n.b(L_SIGMA), rows_natural(1, 1)});
}

void add_static_head(const Net& n, Stream& st, int x6 = 0) {
    const int W = n.W, W2 = n.W2;
    st.segs.push_back(seg(n.NTW, W / 2, k_natural(W / 2, 0), rows_natural(n.NTW, W), n.w(L_FINAL), W));
    mark(st.segs.back(), x6);                                          // xyz_encoding_final: a 256x256 product like the trunk
    st.bias.push_back({n.b(L_FINAL), rows_natural(n.NTW, W)});
    st.segs.push_back(seg(n.NTH, W / 2, k_natural(W / 2, 0), rows_natural(n.NTH, W2), n.w(L_DIR), W + 27));
    mark(st.segs.back(), x6);
    st.segs.push_back(seg(n.NTH, NEFES_D_STEPS, k_emb(4, NEFES_D_STEPS, W), rows_natural(n.NTH, W2), n.w(L_DIR), W + 27));
    st.bias.push_back({n.b(L_DIR), rows_natural(n.NTH, W2)});
    st.segs.push_back(seg(n.NTR, W2 / 2, k_natural(W2 / 2, 0), rows_natural(n.NTR, 3 + n.C), n.w(L_RGB), W2));
    st.bias.push_back({n.b(L_RGB), rows_natural(n.NTR, 3 + n.C)});
}

// the same head with every product as an fp16 two-part split product (NEFES_STREAM_FWD_STATIC_H3): the segment order is the
// first NEFES_H3F_N_STATIC segments of the full fp16 stream, with dir_encoding alone where that one stacks [dir ; t0]
void add_static_head_h3(const Net& n, Stream& st) {
    const int W = n.W, W2 = n.W2;
    st.segs.push_back(seg(n.NTW, W / 2, k_natural(W / 2, 0), rows_natural(n.NTW, W), n.w(L_FINAL), W));
    mark(st.segs.back(), 2);
    st.bias.push_back({n.b(L_FINAL), rows_natural(n.NTW, W)});
    st.segs.push_back(seg(n.NTH, W / 2, k_natural(W / 2, 0), rows_natural(n.NTH, W2), n.w(L_DIR), W + 27));
    mark(st.segs.back(), 2);
    st.segs.push_back(seg(n.NTH, 16, k_emb(4, 16, W), rows_natural(n.NTH, W2), n.w(L_DIR), W + 27));   // 14 k-steps padded to 16
    mark(st.segs.back(), 2);
    st.bias.push_back({n.b(L_DIR), rows_natural(n.NTH, W2)});
    st.segs.push_back(seg(n.NTR, W2 / 2, k_natural(W2 / 2, 0), rows_natural(n.NTR, 3 + n.C), n.w(L_RGB), W2));
    mark(st.segs.back(), 2);
    st.bias.push_back({n.b(L_RGB), rows_natural(n.NTR, 3 + n.C)});
}

// x6 forward streams: dir_encoding and transient_encoding.0 read the same input (cat[final, dir-emb]), so they run as ONE
// product with 2*NTH tiles (rows: dir | t0): one operand split instead of two, and the standard 8-tile shape at Wd = 256.
// Segment order: FINAL, [DIR;T0] hidden part, [DIR;T0] direction part, RGB, T1, T2, TH (all x6); bias blocks keep the
// order of the fp32 streams (FINAL, DIR, RGB, T0, T1, T2, TH).
void add_heads_x6(const Net& n, Stream& st, int kind = 1) {
    const int W = n.W, W2 = n.W2;
    st.segs.push_back(seg(n.NTW, W / 2, k_natural(W / 2, 0), rows_natural(n.NTW, W), n.w(L_FINAL), W));
    mark(st.segs.back(), kind);
    st.bias.push_back({n.b(L_FINAL), rows_natural(n.NTW, W)});
    st.segs.push_back(seg(2 * n.NTH, W / 2, k_natural(W / 2, 0), rows_natural(2 * n.NTH, 2 * W2), n.dt_w.data(), W + 27));
    mark(st.segs.back(), kind);
    // direction part: 14 k-steps padded to 16 (= two 16-k steps; slots 14, 15 are padding)
    st.segs.push_back(seg(2 * n.NTH, 16, k_emb(4, 16, W), rows_natural(2 * n.NTH, 2 * W2), n.dt_w.data(), W + 27));
    mark(st.segs.back(), kind);
    st.bias.push_back({n.b(L_DIR), rows_natural(n.NTH, W2)});
    st.segs.push_back(seg(n.NTR, W2 / 2, k_natural(W2 / 2, 0), rows_natural(n.NTR, 3 + n.C), n.w(L_RGB), W2));
    mark(st.segs.back(), kind);
    st.bias.push_back({n.b(L_RGB), rows_natural(n.NTR, 3 + n.C)});
    st.bias.push_back({n.b(L_T0), rows_natural(n.NTH, W2)});
    for (int l = L_T1; l <= L_T2; ++l) {
        st.segs.push_back(seg(n.NTH, W2 / 2, k_natural(W2 / 2, 0), rows_natural(n.NTH, W2), n.w(l), W2));
        mark(st.segs.back(), kind);
        st.bias.push_back({n.b(l), rows_natural(n.NTH, W2)});
    }
    st.segs.push_back(seg(1, W2 / 2, k_natural(W2 / 2, 0), rows_natural(1, 5), n.th_w.data(), W2));
    mark(st.segs.back(), kind);
    st.bias.push_back({n.th_b.data(), rows_natural(1, 5)});
}

void add_transient_head(const Net& n, Stream& st, int x6 = 0) {
    const int W = n.W, W2 = n.W2;
    // (transient_encoding.0 stays fp32: with both it and dir_encoding on bf16x6 the forward kernel spills registers)
    st.segs.push_back(seg(n.NTH, W / 2, k_natural(W / 2, 0), rows_natural(n.NTH, W2), n.w(L_T0), W + 27));
    st.segs.push_back(seg(n.NTH, NEFES_D_STEPS, k_emb(4, NEFES_D_STEPS, W), rows_natural(n.NTH, W2), n.w(L_T0), W + 27));
    st.bias.push_back({n.b(L_T0), rows_natural(n.NTH, W2)});
    for (int l = L_T1; l <= L_T2; ++l) {
        st.segs.push_back(seg(n.NTH, W2 / 2, k_natural(W2 / 2, 0), rows_natural(n.NTH, W2), n.w(l), W2));
        mark(st.segs.back(), x6);
        st.bias.push_back({n.b(l), rows_natural(n.NTH, W2)});
    }
    st.segs.push_back(seg(1, W2 / 2, k_natural(W2 / 2, 0), rows_natural(1, 5), n.th_w.data(), W2));
    st.bias.push_back({n.th_b.data(), rows_natural(1, 5)});
}

// backward-to-inputs stream: A operand = W^T, B operand = upstream gradient vector
void add_backward(const Net& n, Stream& st, int x6 = 0, bool transient = true) {
    const int W = n.W, W2 = n.W2, NTW = n.NTW, NTH = n.NTH;
    // static rgb/feature head^T first (its 3+C upstream values are consumed straight after the tile's loads):
    // in = 3+C grads (compact slots), out = d g
    if (x6 == 2) {   // fp16 stream: natural slots, the head class's k-steps of 16 (channels past 3+C are zero columns)
        const int ks = 8 * nefes_head_kr16(n.C);
        std::vector<int> k = k_natural(ks, 0);
        for (auto& v : k) if (v >= 3 + n.C) v = -1;
        st.segs.push_back(seg(NTH, ks, k, rows_natural(NTH, W2), n.w(L_RGB), W2, true));
        mark(st.segs.back(), 2);
    } else {
        const int kr = (3 + n.C + 1) / 2;
        st.segs.push_back(seg(NTH, kr, k_compact(kr, 3 + n.C), rows_natural(NTH, W2), n.w(L_RGB), W2, true));
    }
    if (transient) {
    // transient heads^T: in = 5 pre-activation grads (compact slots), out = d t2
    st.segs.push_back(seg(NTH, 3, k_compact(3, 5), rows_natural(NTH, W2), n.th_w.data(), W2, true));
    st.segs.push_back(seg(NTH, W2 / 2, k_natural(W2 / 2, 0), rows_natural(NTH, W2), n.w(L_T2), W2, true));
    mark(st.segs.back(), x6);
    st.segs.push_back(seg(NTH, W2 / 2, k_natural(W2 / 2, 0), rows_natural(NTH, W2), n.w(L_T1), W2, true));
    mark(st.segs.back(), x6);
    }
    // [transient_encoding.0 ; dir_encoding]^T: out rows = dir-embedding slots (1 tile) then final features (NTW tiles)
    std::vector<int> rows_fd = concat(rows_emb(4, 1, W), rows_natural(NTW, W));
    if (transient) {
        st.segs.push_back(seg(NTW + 1, W2 / 2, k_natural(W2 / 2, 0), rows_fd, n.w(L_T0), W + 27, true));
        mark(st.segs.back(), x6);
    }
    st.segs.push_back(seg(NTW + 1, W2 / 2, k_natural(W2 / 2, 0), rows_fd, n.w(L_DIR), W + 27, true));
    mark(st.segs.back(), x6);
    // xyz_encoding_final^T, plus the static-sigma head as one extra k-step
    st.segs.push_back(seg(NTW, W / 2, k_natural(W / 2, 0), rows_natural(NTW, W), n.w(L_FINAL), W, true));
    mark(st.segs.back(), x6);
    st.segs.push_back(seg(NTW, 1, k_compact(1, 1), rows_natural(NTW, W), n.w(L_SIGMA), W, true));
    for (int l = 7; l >= 0; --l) {
        if (l == 4) {
            // d xyz-embedding tiles first (2 tiles of frequency slots, or 1 tile for an external embedding), then d hidden
            const std::vector<int> re = rows_xyz(n, 0);
            std::vector<int> rows = concat(re, rows_natural(NTW, W));
            for (int i = (int)re.size(); i < (int)rows.size(); ++i) rows[i] += n.in_xyz;
            st.segs.push_back(seg(NTW + (int)re.size() / 32, W / 2, k_natural(W / 2, 0), rows, n.w(4), n.in_xyz + W, true));
            mark(st.segs.back(), x6);
        } else if (l == 0) {
            const std::vector<int> re = rows_xyz(n, 0);
            st.segs.push_back(seg((int)re.size() / 32, W / 2, k_natural(W / 2, 0), re, n.w(0), n.in_xyz, true));
            mark(st.segs.back(), x6);
        } else {
            st.segs.push_back(seg(NTW, W / 2, k_natural(W / 2, 0), rows_natural(NTW, W), n.w(l), W, true));
            mark(st.segs.back(), x6);
        }
    }
}

// Elements of the weight matrix a segment reads (the whole torch tensor, or one of the virtual stacked matrices).
static size_t matrix_elems(const Net& n, const float* Wm) {
    if (Wm == n.th_w.data()) return n.th_w.size();
    if (Wm == n.dt_w.data()) return n.dt_w.size();
    const int W = n.W, W2 = n.W2;
    for (int l = 0; l < 18; ++l) {
        if (n.t[2 * l] != Wm) continue;
        if (l < 8) return (size_t)W * (l == 0 ? n.in_xyz : (l == 4 ? n.in_xyz + W : W));
        if (l == L_FINAL) return (size_t)W * W;
        if (l == L_DIR || l == L_T0) return (size_t)W2 * (W + 27);
        if (l == L_SIGMA) return (size_t)W;
        if (l == L_RGB) return (size_t)(3 + n.C) * W2;
        if (l == L_T1 || l == L_T2) return (size_t)W2 * W2;
        if (l == L_TRGB) return (size_t)3 * W2;
        return (size_t)W2;
    }
    return 0;
}
static float abs_max(const float* p, size_t cnt) {
    float m = 0.f;
    for (size_t i = 0; i < cnt; ++i) {
        const float a = fabsf(p[i]);
        if (a > m) m = a;
    }
    return m;
}
// fp16 segments: one power-of-two scale per weight MATRIX (all segments reading a matrix -- e.g. the hidden and the xyz part of
// layer 5 -- accumulate into the same tiles and must agree); dir_encoding and transient_encoding.0 count as one matrix (their
// products share accumulators in both directions).
static void assign_weight_exponents(const Net& n, Stream (&st)[NEFES_N_STREAMS]) {
    const int e_dt = n.transient ? scale_exp(abs_max(n.dt_w.data(), n.dt_w.size())) : 0;
    for (auto& stream : st)
        for (auto& sg : stream.segs) {
            if (!sg.h3 || !sg.W) continue;
            const bool dt = sg.W == n.dt_w.data() || (n.transient && (sg.W == n.w(L_DIR) || sg.W == n.w(L_T0)));
            sg.wexp = dt ? e_dt : scale_exp(abs_max(sg.W, matrix_elems(n, sg.W)));
        }
}

// Exponent group of a weight matrix = the layer index whose scale it shares (assign_weight_exponents): its own, except
// transient_encoding.0 -> dir_encoding (transient networks) and the three transient heads -> one stacked matrix.  -1: none.
static int group_of(const Net& n, const float* Wm) {
    if (!Wm) return -1;
    if (Wm == n.th_w.data()) return L_TRGB;
    if (Wm == n.dt_w.data()) return L_DIR;
    for (int l = 0; l < 18; ++l)
        if (n.t[2 * l] == Wm) return (n.transient && l == L_T0) ? (int)L_DIR : ((l == L_TSIGMA || l == L_TBETA) ? (int)L_TRGB : l);
    return -1;
}

bool build(const NefesNetDesc* d, const float* const* tensors, Net& n, Stream (&st)[NEFES_N_STREAMS]) {
    if (!d) return false;
    if (d->width != 128 && d->width != 256) return false;
    if (nefes_head_class(d->feat_dim) < 0) return false;          // 0 <= C <= NEFES_HEAD_MAX_C (layout.h: head classes)
    n.W = d->width; n.W2 = n.W / 2; n.C = d->feat_dim;
    n.NTW = n.W / 32; n.NTH = n.W2 / 32; n.NTR = nefes_head_ntr(n.C);
    n.transient = d->has_transient != 0;
    n.ext = d->xyz_encoding == NEFES_XYZ_EXTERNAL32;
    if (d->xyz_encoding != NEFES_XYZ_FREQ10 && !n.ext) return false;
    n.in_xyz = n.ext ? 32 : 63;
    n.e_steps = n.ext ? NEFES_X_STEPS : NEFES_E_STEPS;
    n.t = tensors;
    if (tensors && n.transient) {
        n.th_w.assign((size_t)5 * n.W2, 0.f);
        n.th_b.assign(5, 0.f);
        memcpy(&n.th_w[0], n.w(L_TRGB), sizeof(float) * 3 * n.W2);
        memcpy(&n.th_w[(size_t)3 * n.W2], n.w(L_TSIGMA), sizeof(float) * n.W2);
        memcpy(&n.th_w[(size_t)4 * n.W2], n.w(L_TBETA), sizeof(float) * n.W2);
        memcpy(&n.th_b[0], n.b(L_TRGB), sizeof(float) * 3);
        n.th_b[3] = n.b(L_TSIGMA)[0];
        n.th_b[4] = n.b(L_TBETA)[0];
        n.dt_w.resize((size_t)2 * n.W2 * (n.W + 27));
        memcpy(&n.dt_w[0], n.w(L_DIR), sizeof(float) * n.W2 * (n.W + 27));
        memcpy(&n.dt_w[(size_t)n.W2 * (n.W + 27)], n.w(L_T0), sizeof(float) * n.W2 * (n.W + 27));
    } else {
        n.dt_w.assign((size_t)2 * n.W2 * (n.W + 27), 0.f);
        n.th_w.assign((size_t)5 * n.W2, 0.f);
        n.th_b.assign(5, 0.f);
    }
    // a null `tensors` is allowed for geometry queries: substitute a dummy table
    static const float* dummy[36] = {nullptr};
    if (!tensors) n.t = dummy;
    add_trunk(n, st[NEFES_STREAM_FWD_SIGMA]);
    add_trunk(n, st[NEFES_STREAM_FWD_STATIC]);
    add_static_head(n, st[NEFES_STREAM_FWD_STATIC]);
    add_backward(n, st[NEFES_STREAM_BWD_STATIC], false, false);
    if (n.transient) {
        add_trunk(n, st[NEFES_STREAM_FWD_FULL]);
        add_static_head(n, st[NEFES_STREAM_FWD_FULL]);
        add_transient_head(n, st[NEFES_STREAM_FWD_FULL]);
        add_backward(n, st[NEFES_STREAM_BWD_FULL]);
    }
    // bf16x6 instances exist for the two canonical shapes only (and the sigma-only pass at every Wd = 256 network); the fp16 two-part
    // instances for both widths x both head classes (frequency embedding), and for Wd = 256 / class 0 with an external embedding
    const bool big = n.W == 256, small = n.W == 128 && n.C == 128 && !n.ext;
    const bool h3_shape = !n.ext || (n.W == 256 && nefes_head_class(n.C) == 0);
    if (big || small) add_trunk(n, st[NEFES_STREAM_FWD_SIGMA_X6], 1);
    if (n.transient && (big || small) && (small || n.C == 16)) {
        add_trunk(n, st[NEFES_STREAM_FWD_FULL_X6], 1);
        add_heads_x6(n, st[NEFES_STREAM_FWD_FULL_X6]);
        add_backward(n, st[NEFES_STREAM_BWD_FULL_X6], 1);
    }
    if (h3_shape) {
        add_trunk(n, st[NEFES_STREAM_FWD_SIGMA_H3], 2);
        st[NEFES_STREAM_FWD_SIGMA_H3].h3 = true;
        if (!n.ext) {                                   // static head only: what a coarse network runs in train mode
            add_trunk(n, st[NEFES_STREAM_FWD_STATIC_H3], 2);
            add_static_head_h3(n, st[NEFES_STREAM_FWD_STATIC_H3]);
            add_backward(n, st[NEFES_STREAM_BWD_STATIC_H3], 2, false);
            st[NEFES_STREAM_FWD_STATIC_H3].h3 = st[NEFES_STREAM_BWD_STATIC_H3].h3 = true;
        }
        if (n.transient) {
            add_trunk(n, st[NEFES_STREAM_FWD_FULL_H3], 2);
            add_heads_x6(n, st[NEFES_STREAM_FWD_FULL_H3], 2);
            add_backward(n, st[NEFES_STREAM_BWD_FULL_H3], 2);
            st[NEFES_STREAM_FWD_FULL_H3].h3 = st[NEFES_STREAM_BWD_FULL_H3].h3 = true;
        }
    }
    if (tensors) assign_weight_exponents(n, st);
    return true;
}

const uint64_t kHeaderBytes = NEFES_BLOB_HEADER_BYTES;
static_assert(sizeof(NefesBlobInfo) <= NEFES_BLOB_HEADER_BYTES, "blob header too small");
uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

void fill_info(const Stream (&st)[NEFES_N_STREAMS], int width, NefesBlobInfo* info) {
    uint64_t off = kHeaderBytes;
    for (int k = 0; k < NEFES_N_STREAMS; ++k) {
        NefesStreamInfo& si = info->stream[k];
        const int kib = nefes_stream_slab_kib(k, width);
        si.n_slabs = (uint32_t)st[k].n_slabs(NEFES_FRAGS_OF_KIB(kib));
        si.bias_floats = (uint32_t)st[k].bias_floats();
        si.scale_off = (uint32_t)st[k].bias_only();
        si.scale_count = (uint32_t)st[k].scale_count();
        if (si.n_slabs == 0) { si.slab_off = si.bias_off = 0; si.bias_floats = si.scale_off = si.scale_count = 0; continue; }
        si.bias_off = off;
        off = align_up(off + 4ull * si.bias_floats, 256);
        si.slab_off = off;
        off += (uint64_t)si.n_slabs * kib * 1024;
    }
    info->total_bytes = off;
}

}  // namespace

extern "C" int nefes_version(void) { return NEFES_ABI_VERSION; }

extern "C" size_t nefes_stream_slab_bytes(const NefesNetDesc* desc, int stream) {
    if (!desc || stream < 0 || stream >= 13) return 0;
    return (size_t)nefes_stream_slab_kib(stream, desc->width) * 1024;
}

extern "C" int nefes_blob_info(const NefesNetDesc* desc, NefesBlobInfo* info) {
    if (!desc || !info) return NEFES_E_BADARG;
    Net n;
    Stream st[NEFES_N_STREAMS];
    if (!build(desc, nullptr, n, st)) return NEFES_E_UNSUPPORTED;
    fill_info(st, desc->width, info);
    return 0;
}

// Number of elements of tensor i of the (weight, bias) table for this description (torch shapes, nerfh_nff.py:452-505).
static int64_t tensor_elems(const Net& n, int i) {
    const int l = i / 2;
    const bool bias = i & 1;
    int out, in;
    if (l < 8) { out = n.W; in = l == 0 ? n.in_xyz : (l == 4 ? n.in_xyz + n.W : n.W); }
    else if (l == L_FINAL) { out = n.W; in = n.W; }
    else if (l == L_DIR || l == L_T0) { out = n.W2; in = n.W + 27; }
    else if (l == L_SIGMA) { out = 1; in = n.W; }
    else if (l == L_RGB) { out = 3 + n.C; in = n.W2; }
    else if (l == L_T1 || l == L_T2) { out = n.W2; in = n.W2; }
    else if (l == L_TRGB) { out = 3; in = n.W2; }
    else { out = 1; in = n.W2; }    // transient sigma, beta
    return bias ? out : (int64_t)out * in;
}

// One walk over the streams serves both products: `base` != null writes the blob (values); `map` != null writes, for every
// 16-bit slot of the blob, the code (flat parameter index + 1) << 3 | part that nefes_pack_device expands on the GPU
// (part 0/1 = low/high half of the fp32 value; 2/3/4 = the bf16 hi/mid/lo parts of the x6 split; code 0 = zero).  In map
// mode the tensors hold float(flat index + 1), which every copy in build() and at()/at16() carries along unchanged.
static int pack_walk(const NefesNetDesc* desc, const Net& n, char* base, uint32_t* map, const NefesBlobInfo& info,
                     Stream (&st)[NEFES_N_STREAMS]) {
    auto code = [](float v, int part) { return v == 0.f ? 0u : (((uint32_t)v) << 3 | (uint32_t)part); };
    // fp16 two-part slots: the exponent group (bits 27..31) rides along, parts 5 / 6 = hi / lo of value * 2^(group exponent)
    auto code16 = [](float v, int part, int group) { return v == 0.f ? 0u : ((uint32_t)group << 27 | ((uint32_t)v) << 3 | (uint32_t)part); };
    for (int k = 0; k < NEFES_N_STREAMS; ++k) {
        const NefesStreamInfo& si = info.stream[k];
        if (si.n_slabs == 0) continue;
        uint64_t boff = si.bias_off;
        for (auto& bb : st[k].bias)
            for (int r : bb.ridx) {
                const float v = r < 0 ? 0.f : bb.b[r];
                if (base) memcpy(base + boff, &v, 4);
                if (map) { map[boff / 2] = code(v, 0); map[boff / 2 + 1] = code(v, 1); }
                boff += 4;
            }
        if (st[k].h3) {   // scale table (layout.h): per segment (exponent, row bound), then max |b| per bias block; padded to 4 words
            const int ns = (int)st[k].segs.size(), nb = (int)st[k].bias.size();
            for (int i = 0; i < st[k].scale_count(); ++i) {
                uint32_t word = 0;
                if (!map && i < 2 * ns) {
                    const Seg& sg = st[k].segs[i / 2];
                    if (i % 2 == 0) { const int32_t e = sg.wexp; memcpy(&word, &e, 4); }
                    else { const float b = sg.row_bound(); memcpy(&word, &b, 4); }
                } else if (!map && i < 2 * ns + nb) {
                    const BiasBlk& bb = st[k].bias[i - 2 * ns];
                    float m = 0.f;
                    for (int r : bb.ridx)
                        if (r >= 0 && fabsf(bb.b[r]) > m) m = fabsf(bb.b[r]);
                    memcpy(&word, &m, 4);
                }
                if (base) memcpy(base + boff, &word, 4);
                if (map) {   // exponent words: halves 1 / 2 of the group's exponent (part 7); bounds and bias maxima are written by
                             // the device plan's reductions (nefes_pack_h3_plan): the expansion leaves those words alone
                    uint32_t lo = 0, hi = 0;
                    if (i < 2 * ns && i % 2 == 0) {
                        const Seg& sg = st[k].segs[i / 2];
                        const int g = sg.h3 ? group_of(n, sg.W) : -1;
                        if (g >= 0) { lo = (uint32_t)g << 27 | 1u << 3 | 7u; hi = (uint32_t)g << 27 | 2u << 3 | 7u; }
                    } else if (i < 2 * ns + nb) {
                        lo = hi = NEFES_PACK_KEEP;
                    }
                    map[boff / 2] = lo; map[boff / 2 + 1] = hi;
                }
                boff += 4;
            }
        }
        uint64_t soff = si.slab_off;
        const int frags = NEFES_FRAGS_OF_KIB(nefes_stream_slab_kib(k, desc->width));
        const uint64_t slab_bytes = (uint64_t)frags * 256;
        for (auto& sg : st[k].segs) {
            if (sg.h3) {   // units of two 1 KiB groups (hi, lo): lane = 8 fp16 of W 2^wexp = A operand of one 32x32x16 f16 MFMA
                const int ups = (frags / 4) / 2;
                for (int sl = 0; sl < sg.slabs(frags); ++sl, soff += slab_bytes) {
                    for (int uu = 0; uu < ups && sl * ups + uu < sg.units(); ++uu) {
                        const int u = sl * ups + uu, q = u / sg.nt, t = u % sg.nt;
                        const uint64_t grp = soff + (uint64_t)uu * 2048;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 8; ++i) {
                                const float v = sg.at16(q, i, t, lane);
                                uint16_t part[2] = {0, 0};
                                if (!map) split_f16x2(v, sg.wexp, part);
                                for (int pp = 0; pp < 2; ++pp) {
                                    const uint64_t o = grp + 2ull * (pp * 512 + lane * 8 + i);
                                    if (base) memcpy(base + o, &part[pp], 2);
                                    if (map) map[o / 2] = code16(v, 5 + pp, group_of(n, sg.W));
                                }
                            }
                    }
                }
                continue;
            }
            if (sg.x6) {   // units of three 1 KiB groups (hi, mid, lo): lane = 8 bf16 = A operand of one 32x32x16 MFMA
                const int ups = (frags / 4) / 3;
                for (int sl = 0; sl < sg.slabs(frags); ++sl, soff += slab_bytes) {
                    for (int uu = 0; uu < ups && sl * ups + uu < sg.units(); ++uu) {
                        const int u = sl * ups + uu, q = u / sg.nt, t = u % sg.nt;
                        const uint64_t grp = soff + (uint64_t)uu * 3072;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 8; ++i) {
                                const float v = sg.at16(q, i, t, lane);
                                uint16_t part[3];
                                split_bf16x3(v, part);
                                for (int pp = 0; pp < 3; ++pp) {
                                    const uint64_t o = grp + 2ull * (pp * 512 + lane * 8 + i);
                                    if (base) memcpy(base + o, &part[pp], 2);
                                    if (map) map[o / 2] = code(v, 2 + pp);
                                }
                            }
                    }
                }
                continue;
            }
            const int sps = nefes_steps_per_slab(sg.nt, frags);
            for (int sl = 0; sl < sg.slabs(frags); ++sl, soff += slab_bytes) {
                const int steps = (sg.ks - sl * sps) < sps ? (sg.ks - sl * sps) : sps;
                for (int f = 0; f < steps * sg.nt; ++f) {
                    const int s = sl * sps + f / sg.nt, t = f % sg.nt;
                    const uint64_t dst = soff + 4ull * ((f >> 2) * 256 + (f & 3));   // b128 group f/4, component f%4
                    for (int lane = 0; lane < 64; ++lane) {
                        const float v = sg.at(s, t, lane);
                        const uint64_t o = dst + 16ull * lane;
                        if (base) memcpy(base + o, &v, 4);
                        if (map) { map[o / 2] = code(v, 0); map[o / 2 + 1] = code(v, 1); }
                    }
                }
            }
        }
    }
    return 0;
}

extern "C" int nefes_pack_weights(const NefesNetDesc* desc, const float* const* tensors, int n_tensors, void* blob,
                                  size_t blob_bytes) {
    if (!desc || !tensors || !blob) return NEFES_E_BADARG;
    const int need = desc->has_transient ? 36 : 24;
    if (n_tensors < need) return NEFES_E_BADARG;
    for (int i = 0; i < need; ++i)
        if (!tensors[i]) return NEFES_E_BADARG;
    Net n;
    Stream st[NEFES_N_STREAMS];
    if (!build(desc, tensors, n, st)) return NEFES_E_UNSUPPORTED;
    NefesBlobInfo info;
    fill_info(st, desc->width, &info);
    if (blob_bytes < info.total_bytes) return NEFES_E_BADBLOB;
    char* base = (char*)blob;
    memset(base, 0, info.total_bytes);
    uint32_t* hdr = (uint32_t*)base;
    hdr[0] = 0x5346454eu;  // 'NEFS'
    hdr[1] = NEFES_ABI_VERSION;
    memcpy(hdr + 2, desc, sizeof(*desc));
    memcpy(hdr + 8, &info, sizeof(info));
    return pack_walk(desc, n, base, nullptr, info, st);
}

// The network built over index tensors: tensor i holds float(flat index + 1) of its elements (flat = the tensor table
// concatenated in order), which every copy in build() and at()/at16() carries along unchanged.
struct Indexed {
    std::vector<std::vector<float>> idx;
    const float* ptrs[36] = {nullptr};
    int64_t off[37] = {0};
    int need = 0;
    Net n;
    Stream st[NEFES_N_STREAMS];
    NefesBlobInfo info;
};
static int build_indexed(const NefesNetDesc* desc, Indexed& x, int64_t* tensor_elems_out) {
    {   // geometry first (tensor sizes depend on it)
        Stream tmp[NEFES_N_STREAMS];
        if (!build(desc, nullptr, x.n, tmp)) return NEFES_E_UNSUPPORTED;
    }
    x.need = desc->has_transient ? 36 : 24;
    x.idx.resize(x.need);
    int64_t flat = 0;
    for (int i = 0; i < x.need; ++i) {
        const int64_t ne = tensor_elems(x.n, i);
        if (tensor_elems_out) tensor_elems_out[i] = ne;
        x.idx[i].resize((size_t)ne);
        for (int64_t e = 0; e < ne; ++e) x.idx[i][(size_t)e] = (float)(flat + e + 1);
        x.off[i] = flat;
        flat += ne;
        x.ptrs[i] = x.idx[i].data();
    }
    x.off[x.need] = flat;
    if (flat + 1 >= (1 << 24)) return NEFES_E_UNSUPPORTED;   // codes are carried as exact fp32 integers
    if (!build(desc, x.ptrs, x.n, x.st)) return NEFES_E_UNSUPPORTED;   // (x is used in place: the segments point into x.n)
    fill_info(x.st, desc->width, &x.info);
    return 0;
}

extern "C" int nefes_pack_map(const NefesNetDesc* desc, uint32_t* map, size_t n_entries, int64_t* tensor_elems_out) {
    if (!desc || !map) return NEFES_E_BADARG;
    Indexed x;
    const int rc = build_indexed(desc, x, tensor_elems_out);
    if (rc) return rc;
    if (n_entries < x.info.total_bytes / 2) return NEFES_E_BADBLOB;
    memset(map, 0, sizeof(uint32_t) * (x.info.total_bytes / 2));
    return pack_walk(desc, x.n, nullptr, map, x.info, x.st);
}

// Reductions the device re-pack of the fp16 two-part streams needs before the slot expansion (pack_device.hip runs one
// workgroup per job): plan = [n_jobs, 0, jobs (8 ints each) ..., index pool ...]
//   kind 0  exponent of group `out`      = scale_exp(max |w|) over up to three flat ranges (offset, count)
//   kind 1  row bound -> blob word `out` : rows a, k-values b, flat index of (row r, k) = pool[c + r] + pool[d + k]
//   kind 2  max |b|   -> blob word `out` : over up to three flat ranges
extern "C" int nefes_pack_h3_plan(const NefesNetDesc* desc, int32_t* plan, size_t n_ints, size_t* needed) {
    if (!desc) return NEFES_E_BADARG;
    Indexed x;
    const int rc = build_indexed(desc, x, nullptr);
    if (rc) return rc;
    const Net& n = x.n;
    std::vector<int32_t> jobs, pool;
    auto job = [&](int kind, int out, const int (&arg)[6]) {
        jobs.push_back(kind); jobs.push_back(out);
        for (int v : arg) jobs.push_back(v);
    };
    const int n_layers = x.need / 2;
    for (int g = 0; g < n_layers; ++g) {
        if (group_of(n, n.t[2 * g]) != g) continue;
        int arg[6] = {0, 0, 0, 0, 0, 0}, cnt = 0;
        for (int l = 0; l < n_layers && cnt < 3; ++l)
            if (group_of(n, n.t[2 * l]) == g) { arg[2 * cnt] = (int)x.off[2 * l]; arg[2 * cnt + 1] = (int)(x.off[2 * l + 1] - x.off[2 * l]); ++cnt; }
        job(0, g, arg);
    }
    auto tensor_range = [&](const float* p, int& o, int& c) {
        for (int i = 0; i < x.need; ++i)
            if (n.t[i] == p) { o = (int)x.off[i]; c = (int)(x.off[i + 1] - x.off[i]); return true; }
        return false;
    };
    for (int k = 0; k < NEFES_N_STREAMS; ++k) {
        const NefesStreamInfo& si = x.info.stream[k];
        if (!x.st[k].h3 || si.n_slabs == 0) continue;
        const int ns = (int)x.st[k].segs.size(), nb = (int)x.st[k].bias.size();
        const int tab = (int)(si.bias_off / 4) + (int)si.scale_off;
        for (int i = 0; i < ns; ++i) {
            const Seg& sg = x.st[k].segs[i];
            int arg[6] = {0, 0, (int)pool.size(), 0, 0, 0};
            std::vector<int> rows, ks;
            for (int r : sg.ridx) if (r >= 0) rows.push_back(r);
            for (int kk : sg.kidx) if (kk >= 0) ks.push_back(kk);
            for (int r : rows) pool.push_back(sg.transposed ? r : (int)sg.W[(size_t)r * sg.ld] - 1);
            arg[3] = (int)pool.size();
            for (int kk : ks) pool.push_back(sg.transposed ? (int)sg.W[(size_t)kk * sg.ld] - 1 : kk);
            arg[0] = (int)rows.size(); arg[1] = (int)ks.size();
            job(1, tab + 2 * i + 1, arg);
        }
        for (int j = 0; j < nb; ++j) {
            const BiasBlk& bb = x.st[k].bias[j];
            int arg[6] = {0, 0, 0, 0, 0, 0};
            if (bb.b == n.th_b.data()) {
                tensor_range(n.b(L_TRGB), arg[0], arg[1]);
                tensor_range(n.b(L_TSIGMA), arg[2], arg[3]);
                tensor_range(n.b(L_TBETA), arg[4], arg[5]);
            } else if (!tensor_range(bb.b, arg[0], arg[1])) {
                return NEFES_E_UNSUPPORTED;
            }
            job(2, tab + 2 * ns + j, arg);
        }
    }
    const int n_jobs = (int)jobs.size() / 8, head = 2 + (int)jobs.size();
    const size_t total = (size_t)head + pool.size();
    if (needed) *needed = total;
    if (!plan) return 0;
    if (n_ints < total) return NEFES_E_BADBLOB;
    plan[0] = n_jobs; plan[1] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        for (int f = 0; f < 8; ++f) plan[2 + 8 * j + f] = jobs[8 * j + f];
        if (jobs[8 * j] == 1) { plan[2 + 8 * j + 4] += head; plan[2 + 8 * j + 5] += head; }   // pool offsets -> absolute
    }
    for (size_t i = 0; i < pool.size(); ++i) plan[head + i] = pool[i];
    return 0;
}
