"""Host-side checks that need no GPU:
 * libnefes_hip.so loads and exports every symbol include/nefes_hip.h declares;
 * the packed weight streams, consumed in the exact order the kernels consume them
   (numpy emulation of the 32x32x2 MFMA lane maps, slabs, slots and tiles of
   nefes_amd/csrc/layout.h), reproduce the oracle's MLP forward and its backward-to-inputs.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from nefes_amd import lib as L
from oracle import ref_cpu as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "nefes_hip.h")).read()
    declared = set(re.findall(r"\b(nefes_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    m = re.search(r"#define NEFES_ABI_VERSION (\d+)", hdr)
    assert lib.nefes_version() == L.ABI_VERSION == int(m.group(1))


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libnefes_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        L.load()


# ---- layout.h restated ------------------------------------------------------------------------
def rho(h, r):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def emb_slot(Lf, s, h):
    if s < 3 * Lf:
        return 3 + 6 * (s // 3) + 3 * h + (s % 3)
    if s == 3 * Lf:
        return h
    if s == 3 * Lf + 1:
        return 2 if h == 0 else -1
    return -1


def emb_vector(e, Lf, steps):
    """[n, 3+6L] embedding in reference order -> slot vector [steps, 2, n]."""
    out = np.zeros((steps, 2, e.shape[0]), np.float32)
    for s in range(steps):
        for h in range(2):
            k = emb_slot(Lf, s, h)
            if k >= 0:
                out[s, h] = e[:, k]
    return out


def emb_vector_T(g, Lf, n_ref):
    """slot vector [steps,2,n] -> [n, n_ref] in reference order (transpose of emb_vector)."""
    out = np.zeros((g.shape[2], n_ref), np.float32)
    for s in range(g.shape[0]):
        for h in range(2):
            k = emb_slot(Lf, s, h)
            if k >= 0:
                out[:, k] = g[s, h]
    return out


def acc_to_vec(acc, t0=0, nt=None):
    nt = acc.shape[0] - t0 if nt is None else nt
    v = np.zeros((nt * 16, 2, acc.shape[2]), np.float32)
    for t in range(nt):
        for r in range(16):
            for h in range(2):
                v[t * 16 + r, h] = acc[t0 + t, rho(h, r)]
    return v


def _slab_kib(which):
    txt = open(os.path.join(os.path.dirname(__file__), '..', 'nefes_amd', 'csrc', 'layout.h')).read()
    return int(re.search(r'#define NEFES_%s_SLAB_KIB (\d+)' % which, txt).group(1))


def _slab_frags(which):
    """256-byte fragments per slab of the forward ('FWD') or backward ('BWD') streams (nefes_amd/csrc/layout.h)."""
    txt = open(os.path.join(os.path.dirname(__file__), '..', 'nefes_amd', 'csrc', 'layout.h')).read()
    return int(re.search(r'#define NEFES_%s_SLAB_KIB (\d+)' % which, txt).group(1)) * 4


class Stream:
    def __init__(self, blob, si, which="FWD"):
        SLAB_FRAGS = self.frags = _slab_frags(which)
        self.slabs = np.frombuffer(blob, np.float32, count=si.n_slabs * SLAB_FRAGS * 64, offset=si.slab_off).reshape(si.n_slabs, SLAB_FRAGS // 4, 64, 4)
        self.bias = np.frombuffer(blob, np.float32, count=si.bias_floats, offset=si.bias_off) if si.bias_floats else None
        self.pos = 0
        self.bpos = 0

    def bias_tiles(self, nt, n):
        b = self.bias[self.bpos:self.bpos + nt * 32].reshape(nt, 32)
        self.bpos += nt * 32
        return np.repeat(b[:, :, None], n, 2).astype(np.float32).copy()

    def mma(self, nt, vec, acc):
        ks = vec.shape[0]
        sps = self.frags // nt
        for sl in range((ks + sps - 1) // sps):
            slab = self.slabs[self.pos]
            self.pos += 1
            steps = min(sps, ks - sl * sps)
            for f in range(steps * nt):
                s, t = sl * sps + f // nt, f % nt
                a = slab[f // 4, :, f % 4]                       # fragment: lane = i + 32*h
                acc[t] += a[:32, None] * vec[s, 0][None, :]       # k = slot(s,0)
                acc[t] += a[32:, None] * vec[s, 1][None, :]       # k = slot(s,1)


def pack(Wd, Cf, typ):
    lib = L.load()
    p = O.make_field_params(typ, Wd, Cf)
    d = L.NefesNetDesc(Wd, Cf, 1 if typ == "fine" else 0, 0)
    info = L.NefesBlobInfo()
    assert lib.nefes_blob_info(d, info) == 0
    names = [n for n, _, _ in O.field_param_shapes(typ, Wd, Cf)]
    arrs = []
    for n in names:
        arrs += [np.ascontiguousarray(p[n + ".weight"].numpy()), np.ascontiguousarray(p[n + ".bias"].numpy())]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    blob = np.zeros(info.total_bytes, np.uint8)
    assert lib.nefes_pack_weights(d, ptrs, len(arrs), blob.ctypes.data, blob.nbytes) == 0
    return p, info, blob.tobytes()


def softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def emulate_forward(info, blob, Wd, Cf, e63, e27, mode):
    """Mirrors field_fwd_kernel<W,NTR,MODE> (csrc/field_fwd.hip) segment by segment."""
    n = e63.shape[0]
    NTW, NTH, NTR = Wd // 32, Wd // 64, (3 + Cf + 31) // 32
    st = Stream(blob, info.stream[{0: L.STREAM_FWD_SIGMA, 1: L.STREAM_FWD_STATIC, 2: L.STREAM_FWD_FULL}[mode]])
    E, D = emb_vector(e63, 10, 32), emb_vector(e27, 4, 14)
    masks = {}
    acc = st.bias_tiles(NTW, n)
    st.mma(NTW, E, acc)
    masks["L1"] = acc > 0
    H = acc_to_vec(np.maximum(acc, 0))
    biases = [st.bias_tiles(NTW, n) for _ in range(7)]          # L2..L8 (bias block order)
    b_sig = st.bias_tiles(1, n)
    out = {}

    def sigma_head():
        sg = b_sig.copy()
        st.mma(1, H, sg)
        out["sigma"] = softplus(sg[0, 0])

    last = 8 if mode == 0 else 9
    b_final = st.bias_tiles(NTW, n) if mode != 0 else None
    for l in range(2, last + 1):
        if mode != 0 and l == 9:
            sigma_head()
        acc = biases[l - 2].copy() if l <= 8 else b_final.copy()
        st.mma(NTW, H, acc)
        if l == 5:
            st.mma(NTW, E, acc)           # skip layer: hidden part first, then the xyz-embedding part
        if l <= 8:
            masks[f"L{l}"] = acc > 0
            acc = np.maximum(acc, 0)
        H = acc_to_vec(acc)
    if mode == 0:
        sigma_head()
        assert st.pos == st.slabs.shape[0]
        return out, masks
    a2 = st.bias_tiles(NTH, n)
    st.mma(NTH, H, a2)
    st.mma(NTH, D, a2)
    masks["DIR"] = a2 > 0
    G = acc_to_vec(np.maximum(a2, 0))
    ar = st.bias_tiles(NTR, n)
    st.mma(NTR, G, ar)
    out["rgbfeat"] = ar.reshape(NTR * 32, n)[:3 + Cf].T
    if mode == 2:
        a2 = st.bias_tiles(NTH, n)
        st.mma(NTH, H, a2)
        st.mma(NTH, D, a2)
        masks["T0"] = a2 > 0
        G = acc_to_vec(np.maximum(a2, 0))
        for tl in (1, 2):
            a2 = st.bias_tiles(NTH, n)
            st.mma(NTH, G, a2)
            masks[f"T{tl}"] = a2 > 0
            G = acc_to_vec(np.maximum(a2, 0))
        th = st.bias_tiles(1, n)
        st.mma(1, G, th)
        sig = lambda x: 1 / (1 + np.exp(-x))
        out["t_rgb"] = sig(th[0, :3]).T
        out["t_sigma"] = softplus(th[0, 3])
        out["t_beta"] = softplus(th[0, 4])
    assert st.pos == st.slabs.shape[0]
    return out, masks


def compact(vals, steps):
    """list of per-sample arrays -> compact slot vector (slot index 2s+h)."""
    n = vals[0].shape[0]
    v = np.zeros((steps, 2, n), np.float32)
    for k, a in enumerate(vals):
        v[k // 2, k % 2] = a
    return v


class MixedStream:
    """Backward bf16x6 stream: fp32 segments (fragments of 256 B) and x6 segments (3 KiB units) in one slab sequence."""

    def __init__(self, blob, si, slab_kib):
        self.f32 = Stream(blob, si, "BWD")
        self.x6 = StreamX6(blob, si, slab_kib)

    def mma(self, nt, vec, acc):                    # fp32 segment
        self.f32.pos = self.x6.pos
        self.f32.mma(nt, vec, acc)
        self.x6.pos = self.f32.pos

    def mma16(self, nt, vec, acc):                  # bf16x6 segment
        a64 = acc.astype(np.float64)
        self.x6.mma(nt, vec, a64)
        acc[...] = a64

    @property
    def pos(self):
        return self.x6.pos

    @property
    def slabs(self):
        return self.f32.slabs


def emulate_backward(info, blob, Wd, Cf, masks, d_pre, x6=False):
    """Mirrors field_bwd_kernel (csrc/field_bwd.hip).  d_pre: pre-activation head gradients.  x6: the X6 instance's stream."""
    n = d_pre["sigma"].shape[0]
    NTW, NTH = Wd // 32, Wd // 64
    if x6:
        st = MixedStream(blob, info.stream[L.STREAM_BWD_FULL_X6], _slab_kib("BWD"))
        big = st.mma16
    else:
        st = Stream(blob, info.stream[L.STREAM_BWD_FULL], "BWD")
        big = st.mma
    Z = lambda nt: np.zeros((nt, 32, n), np.float32)
    C3 = 3 + Cf
    a2 = Z(NTH)
    st.mma(NTH, compact([d_pre["rgbfeat"][:, k] for k in range(C3)], (C3 + 1) // 2), a2)
    Gv = acc_to_vec(a2 * masks["DIR"])
    a2 = Z(NTH)
    st.mma(NTH, compact([d_pre["t_rgb"][:, 0], d_pre["t_rgb"][:, 1], d_pre["t_rgb"][:, 2], d_pre["t_sigma"], d_pre["t_beta"]], 3), a2)
    Tv = acc_to_vec(a2 * masks["T2"])
    for tl in (2, 1):
        a2 = Z(NTH)
        big(NTH, Tv, a2)
        Tv = acc_to_vec(a2 * masks[f"T{tl - 1}"])
    a9 = Z(NTW + 1)
    big(NTW + 1, Tv, a9)
    big(NTW + 1, Gv, a9)
    dD = acc_to_vec(a9, 0, 1)             # tile 0 = d dir-embedding, tiles 1.. = d final
    H = acc_to_vec(a9, 1, NTW)
    acc = Z(NTW)
    big(NTW, H, acc)
    st.mma(NTW, compact([d_pre["sigma"]], 1), acc)
    H = acc_to_vec(acc * masks["L8"])
    accE = Z(2)
    for l in range(8, 1, -1):
        if l == 5:
            a10 = Z(NTW + 2)
            big(NTW + 2, H, a10)
            accE = a10[:2].copy()
            H = acc_to_vec(a10[2:] * masks["L4"])
        else:
            acc = Z(NTW)
            big(NTW, H, acc)
            H = acc_to_vec(acc * masks[f"L{l - 1}"])
    big(2, H, accE)
    assert st.pos == st.slabs.shape[0]
    return emb_vector_T(acc_to_vec(accE), 10, 63), emb_vector_T(dD[:14], 4, 27)


@pytest.mark.parametrize("Wd,Cf", [(256, 16), (128, 128)])
def test_packed_streams_reproduce_the_mlp(Wd, Cf):
    n = 8
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(n, 3, generator=g) - .5) * 5
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    e63, e27 = O.freq_encode(pts, 10), O.freq_encode(dirs, 4)

    # coarse net: sigma-only and static streams
    pc, info_c, blob_c = pack(Wd, Cf, "coarse")
    out, _ = emulate_forward(info_c, blob_c, Wd, Cf, e63.numpy(), e27.numpy(), 0)
    ref = O.field_forward(pc, e63, sigma_only=True)[:, 0].numpy()
    np.testing.assert_allclose(out["sigma"], ref, rtol=2e-5, atol=2e-6)
    out, _ = emulate_forward(info_c, blob_c, Wd, Cf, e63.numpy(), e27.numpy(), 1)
    ref = O.field_forward(pc, torch.cat([e63, e27], 1), output_transient=False).numpy()
    np.testing.assert_allclose(out["rgbfeat"], ref[:, :3 + Cf], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["sigma"], ref[:, 3 + Cf], rtol=2e-5, atol=2e-6)

    # fine net: full forward, then backward-to-inputs against autograd
    pf, info_f, blob_f = pack(Wd, Cf, "fine")
    out, masks = emulate_forward(info_f, blob_f, Wd, Cf, e63.numpy(), e27.numpy(), 2)
    emb = torch.cat([e63, e27], 1).requires_grad_()
    raw = O.field_forward(pf, emb, output_transient=True)
    r = raw.detach().numpy()
    C3 = 3 + Cf
    np.testing.assert_allclose(out["rgbfeat"], r[:, :C3], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["sigma"], r[:, C3], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["t_rgb"], r[:, C3 + 1:C3 + 4], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["t_sigma"], r[:, C3 + 4], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["t_beta"], r[:, C3 + 5], rtol=2e-5, atol=2e-6)

    g_raw = torch.randn(raw.shape, generator=g)
    (g_emb,) = torch.autograd.grad(raw, emb, g_raw)
    gr = g_raw.numpy()
    d_pre = {"rgbfeat": gr[:, :C3], "sigma": gr[:, C3] * (1 - np.exp(-r[:, C3])),
             "t_rgb": gr[:, C3 + 1:C3 + 4] * r[:, C3 + 1:C3 + 4] * (1 - r[:, C3 + 1:C3 + 4]),
             "t_sigma": gr[:, C3 + 4] * (1 - np.exp(-r[:, C3 + 4])), "t_beta": gr[:, C3 + 5] * (1 - np.exp(-r[:, C3 + 5]))}
    scale = np.abs(g_emb.numpy()).max()
    for x6 in ((False, True) if (Wd, Cf) == (256, 16) else (False,)):       # the bf16x6 backward exists for Wd=256, C=16
        g63, g27 = emulate_backward(info_f, blob_f, Wd, Cf, masks, d_pre, x6=x6)
        np.testing.assert_allclose(g63, g_emb.numpy()[:, :63], rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(g27, g_emb.numpy()[:, 63:], rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("Wd,Cf", [(256, 16), (128, 128)])
def test_x6_stream_decodes_to_the_weights(Wd, Cf):
    """bf16x6 streams (layout.h, pack.cpp x6 segments): every weight is stored as an exact (hi, mid, lo) bf16 triple in the
    A-operand order of v_mfma_f32_32x32x16_bf16 -- lane (m, g) of unit (k16-step q, tile t) holds W[32t+m][slot(8q+i, g)].
    Decodes layer 2 of the sigma-only x6 stream (the first hidden 1:1 layer) back into the weight matrix."""
    p, info, blob = pack(Wd, Cf, "coarse")
    si = info.stream[L.STREAM_FWD_SIGMA_X6]
    assert si.n_slabs > 0
    slab_bytes, ups = 48 * 1024, 16
    nt, k16 = Wd // 32, Wd // 16
    # segment order: layer 1 (embedding part: 4 k16-steps x nt tiles), then layer 2
    l1_slabs = (4 * nt + ups - 1) // ups
    raw = np.frombuffer(blob, np.uint16, count=si.n_slabs * slab_bytes // 2, offset=si.slab_off).reshape(si.n_slabs, slab_bytes // 2)
    Wm = p["xyz_encoding_2.0.weight"].numpy()
    got = np.zeros_like(Wm)
    for u in range(k16 * nt):
        sl, uu = l1_slabs + u // ups, u % ups
        q, t = u // nt, u % nt
        unit = raw[sl, uu * 1536:(uu + 1) * 1536].reshape(3, 64, 8).astype(np.uint32)        # [part][lane][i]
        val = sum((unit[pp] << 16).view(np.float32).astype(np.float64) for pp in range(3))     # hi + mid + lo
        for lane in range(64):
            m, g = lane & 31, lane >> 5
            for i in range(8):
                s = 8 * q + i
                got[32 * t + m, 32 * (s >> 4) + rho(g, s & 15)] = val[lane, i]
    assert np.array_equal(got, Wm)                                                           # exact, not approximately


class StreamX6:
    """Consumption of a bf16x6 forward stream (field_x6.h mma_run_x6): units of three 1 KiB groups (hi | mid | lo), 16 units
    per 48 KiB slab, a segment starts on a slab boundary; lane (m, g) of unit (k16-step q, tile t) multiplies slot (8q+i, g)."""

    def __init__(self, blob, si, slab_kib=48):
        self.half = slab_kib * 512
        self.ups = (slab_kib // 3)
        self.raw = np.frombuffer(blob, np.uint16, count=si.n_slabs * self.half, offset=si.slab_off).reshape(si.n_slabs, self.half)
        self.bias = np.frombuffer(blob, np.float32, count=si.bias_floats, offset=si.bias_off)
        self.pos = self.bpos = 0

    def bias_tiles(self, nt, n):
        b = self.bias[self.bpos:self.bpos + nt * 32].reshape(nt, 32)
        self.bpos += nt * 32
        return np.repeat(b[:, :, None], n, 2).astype(np.float64).copy()

    def mma(self, nt, vec, acc):
        k16 = vec.shape[0] // 8
        n_units = k16 * nt
        for u in range(n_units):
            sl, uu = self.pos + u // self.ups, u % self.ups
            q, t = u // nt, u % nt
            unit = self.raw[sl, uu * 1536:(uu + 1) * 1536].reshape(3, 64, 8).astype(np.uint32)
            w = sum((unit[pp] << 16).view(np.float32).astype(np.float64) for pp in range(3))      # [lane][i], exact weights
            for g in range(2):
                acc[t] += w[32 * g:32 * g + 32] @ vec[8 * q:8 * q + 8, g].astype(np.float64)       # [32 rows, 8] @ [8, n]
        self.pos += (n_units + self.ups - 1) // self.ups


@pytest.mark.parametrize("Wd,Cf", [(256, 16), (128, 128)])
def test_x6_sigma_stream_reproduces_the_mlp(Wd, Cf):
    """The whole sigma-only bf16x6 stream consumed in kernel order (field_fwd_x6_kernel<SIGMA>): layer 1 on the embedding,
    layers 2..8 (skip part at layer 5), static_sigma -- against the oracle's forward."""
    n = 8
    g = torch.Generator().manual_seed(6)
    pts = (torch.rand(n, 3, generator=g) - .5) * 5
    e63 = O.freq_encode(pts, 10)
    pc, info, blob = pack(Wd, Cf, "coarse")
    st = StreamX6(blob, info.stream[L.STREAM_FWD_SIGMA_X6])
    NTW = Wd // 32
    E = emb_vector(e63.numpy(), 10, 32)
    b1 = st.bias_tiles(NTW, n)
    biases = [st.bias_tiles(NTW, n) for _ in range(7)]
    b_sig = st.bias_tiles(1, n)
    acc = b1
    st.mma(NTW, E, acc)
    H = acc_to_vec(np.maximum(acc, 0))
    for l in range(2, 9):
        acc = biases[l - 2]
        st.mma(NTW, H, acc)
        if l == 5:
            st.mma(NTW, E, acc)
        H = acc_to_vec(np.maximum(acc, 0))
    sg = b_sig
    st.mma(1, H, sg)
    assert st.pos == st.raw.shape[0]
    ref = O.field_forward(pc, e63, sigma_only=True)[:, 0].numpy()
    np.testing.assert_allclose(softplus(sg[0, 0]), ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("Wd,Cf", [(256, 16), (128, 128)])
def test_x6_full_stream_reproduces_the_mlp(Wd, Cf):
    """The FULL bf16x6 forward stream in kernel order (field_fwd_x6_kernel<FULL>): trunk, static_sigma, xyz_encoding_final,
    the stacked [dir_encoding ; transient_encoding.0] product with its direction part, static_rgb, transient_encoding.2/.4,
    transient heads -- against the oracle's forward (all 3+C+6 raw channels)."""
    n = 8
    g = torch.Generator().manual_seed(7)
    pts = (torch.rand(n, 3, generator=g) - .5) * 5
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    e63, e27 = O.freq_encode(pts, 10), O.freq_encode(dirs, 4)
    pf, info, blob = pack(Wd, Cf, "fine")
    st = StreamX6(blob, info.stream[L.STREAM_FWD_FULL_X6])
    NTW, NTH, NTR = Wd // 32, Wd // 64, (3 + Cf + 31) // 32
    E, D = emb_vector(e63.numpy(), 10, 32), emb_vector(e27.numpy(), 4, 16)       # direction part padded to 16 slots
    b = {"L1": st.bias_tiles(NTW, n)}
    for l in range(2, 9):
        b[f"L{l}"] = st.bias_tiles(NTW, n)
    for name, nt in (("SIG", 1), ("FINAL", NTW), ("DIR", NTH), ("RGB", NTR), ("T0", NTH), ("T1", NTH), ("T2", NTH), ("TH", 1)):
        b[name] = st.bias_tiles(nt, n)
    acc = b["L1"]
    st.mma(NTW, E, acc)
    H = acc_to_vec(np.maximum(acc, 0))
    for l in range(2, 9):
        acc = b[f"L{l}"]
        st.mma(NTW, H, acc)
        if l == 5:
            st.mma(NTW, E, acc)
        H = acc_to_vec(np.maximum(acc, 0))
    sg = b["SIG"]
    st.mma(1, H, sg)
    fin = b["FINAL"]
    st.mma(NTW, H, fin)
    dt = np.concatenate([b["DIR"], b["T0"]], 0)                                      # stacked tiles: dir | t0
    st.mma(2 * NTH, acc_to_vec(fin), dt)
    st.mma(2 * NTH, D, dt)
    ar = b["RGB"]
    st.mma(NTR, acc_to_vec(np.maximum(dt[:NTH], 0)), ar)
    t1 = b["T1"]
    st.mma(NTH, acc_to_vec(np.maximum(dt[NTH:], 0)), t1)
    t2 = b["T2"]
    st.mma(NTH, acc_to_vec(np.maximum(t1, 0)), t2)
    th = b["TH"]
    st.mma(1, acc_to_vec(np.maximum(t2, 0)), th)
    assert st.pos == st.raw.shape[0] and st.bpos == st.bias.shape[0]
    ref = O.field_forward(pf, torch.cat([e63, e27], 1), output_transient=True).numpy()
    sig = lambda x: 1 / (1 + np.exp(-x))
    got = np.concatenate([ar.reshape(NTR * 32, n)[:3 + Cf].T, softplus(sg[0, :1]).T, sig(th[0, :3]).T, softplus(th[0, 3:5]).T], 1)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("Wd,C,tr,enc", [(128, 128, False, 0), (128, 128, True, 0), (256, 16, True, 0), (256, 16, True, 1),
                                         (256, 128, True, 0), (128, 16, False, 0), (256, 77, True, 0)])
def test_pack_map_reproduces_host_pack(Wd, C, tr, enc):
    """nefes_pack_map: expanding the slot -> (parameter, part) codes in numpy exactly as pack_device_kernel does gives the
    blob nefes_pack_weights writes, for every stream (fp32 and bf16x6) and the bias blocks."""
    import ctypes as ct
    from nefes_amd import lib as L
    lib = L.load()
    desc = L.NefesNetDesc(Wd, C, 1 if tr else 0, enc)
    info = L.NefesBlobInfo()
    assert lib.nefes_blob_info(desc, info) == 0
    n = 36 if tr else 24
    elems = (ct.c_int64 * 36)()
    m = np.zeros(info.total_bytes // 2, dtype=np.uint32)
    assert lib.nefes_pack_map(desc, m.ctypes.data_as(ct.c_void_p), m.size, ct.cast(elems, ct.c_void_p)) == 0
    rng = np.random.default_rng(Wd + C + enc)
    tens = [np.ascontiguousarray((rng.standard_normal(s) * 0.1).astype(np.float32)) for s in list(elems)[:n]]
    ptrs = (ct.c_void_p * n)(*[t.ctypes.data for t in tens])
    blob = np.zeros(info.total_bytes, dtype=np.uint8)
    assert lib.nefes_pack_weights(desc, ptrs, n, blob.ctypes.data_as(ct.c_void_p), blob.size) == 0
    flat = np.concatenate([[0.0]] + [t.ravel() for t in tens]).astype(np.float32)
    KEEP = np.uint32(0xffffffff)
    keepw = m == KEEP
    mc = np.where(keepw, 0, m).astype(np.uint32)
    part, field, group = mc & 7, (mc >> 3) & 0xffffff, (mc >> 27).astype(np.int64)
    is_val = (mc != 0) & (part != 7)
    assert int(field[is_val].max()) == flat.size - 1       # every parameter is addressed, none beyond
    x = flat[np.where(is_val, field, 0).astype(np.int64)]

    # ---- the reductions of the fp16 plan (pack_device.hip h3_scales_kernel), in numpy ----
    need = ct.c_size_t(0)
    assert lib.nefes_pack_h3_plan(desc, None, 0, ct.byref(need)) == 0
    plan = np.zeros(need.value, dtype=np.int32)
    assert lib.nefes_pack_h3_plan(desc, plan.ctypes.data_as(ct.c_void_p), plan.size, None) == 0
    f0 = flat[1:]
    gexp = np.zeros(32, dtype=np.int32)
    words = {}

    def ranges_max(j):
        best = np.float32(0)
        for q in range(3):
            off, cnt = int(j[2 + 2 * q]), int(j[3 + 2 * q])
            if cnt:
                best = max(best, np.abs(f0[off:off + cnt]).max())
        return np.float32(best)
    for jb in range(int(plan[0])):
        j = plan[2 + 8 * jb:10 + 8 * jb]
        kind, out = int(j[0]), int(j[1])
        if kind == 0:
            amax = ranges_max(j)
            e = 0
            if amax > 0 and np.isfinite(amax):
                e = int(np.clip(14 + 1 - int(np.frexp(amax)[1]), -60, 60))
            gexp[out] = e
        elif kind == 1:
            A, B = plan[j[4]:j[4] + j[2]].astype(np.int64), plan[j[5]:j[5] + j[3]].astype(np.int64)
            sums = np.cumsum(np.abs(f0[A[:, None] + B[None, :]].astype(np.float64)), axis=1)[:, -1]   # sequential, in double
            best = sums.astype(np.float32).max() if sums.size else np.float32(0)
            words[out] = np.float32(best * np.float32(1.0001))
        else:
            words[out] = ranges_max(j)

    def rne(f):                                           # fp32 -> bf16 bits, round to nearest even
        u = f.view(np.uint32).astype(np.uint64)
        return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    b = x.view(np.uint32)
    hi = rne(x)
    r = (x - (hi << 16).astype(np.uint32).view(np.float32)).astype(np.float32)
    mid = rne(r)
    lo = rne((r - (mid << 16).astype(np.uint32).view(np.float32)).astype(np.float32))
    y = np.ldexp(x, gexp[np.where(part >= 5, group, 0)]).astype(np.float32)              # fp16 two-part: (hi, lo) of x 2^e
    with np.errstate(over="ignore"):
        h_hi = y.astype(np.float16)
        h_lo = (y - h_hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    ge = gexp[group].astype(np.uint32)
    ew = np.where(field == 1, ge & 0xffff, ge >> 16)
    out = np.select([part == 0, part == 1, part == 2, part == 3, part == 4, part == 5, part == 6],
                    [b & 0xffff, b >> 16, hi, mid, lo, h_hi.view(np.uint16).astype(np.uint32), h_lo.view(np.uint16).astype(np.uint32)],
                    ew).astype(np.uint16)
    out[mc == 0] = 0
    out32 = out.view(np.uint32).copy()
    assert np.array_equal(keepw[0::2], keepw[1::2])        # KEEP marks whole 32-bit words
    kw = np.flatnonzero(keepw[0::2])
    assert sorted(words) == sorted(kw.tolist())           # ... exactly the words the plan's reductions write
    for w, v in words.items():
        out32[w] = np.float32(v).view(np.uint32)
    got = out32.view(np.uint16)
    want = blob.view(np.uint16)
    keep = np.ones(m.size, bool)
    keep[:256] = False                                    # 512-byte header
    if not np.array_equal(got[keep], want[keep]):
        bad = np.flatnonzero((got != want) & keep)
        raise AssertionError(f"{bad.size} slots differ, first at byte {2 * int(bad[0])}: got {got[bad[0]]:#x} want {want[bad[0]]:#x} code {m[bad[0]]:#x}")
    assert not m[:256].any()                              # header slots are never written by the device packer
    n_h3 = sum(1 for k in (L.STREAM_FWD_SIGMA_H3, L.STREAM_FWD_FULL_H3, L.STREAM_BWD_FULL_H3, L.STREAM_FWD_STATIC_H3, L.STREAM_BWD_STATIC_H3)
               if info.stream[k].n_slabs)
    assert n_h3 >= (3 if enc else 3 + (2 if tr else 0)) and (part >= 5).any()


def test_m0_only_written_by_the_dma_helper(tmp_path):
    """The LDS-DMA helper (field_common.h lds_dma16) leaves M0 holding the LDS destination instead of saving/restoring it.
    That is sound only while nothing else in the device code reads or writes M0: disassemble every code object of the
    library and check that M0 appears in `s_mov_b32 m0, <sgpr>` and nowhere else."""
    import os, re, subprocess
    from nefes_amd import lib as L
    bindir = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(bindir, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    fat = tmp_path / "fatbin.bin"
    subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", L.LIB_PATH, str(fat)])
    data = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    assert len(starts) >= 8                                  # one bundle per .hip translation unit
    n_dma = 0
    for i, a in enumerate(starts):
        piece = tmp_path / f"bundle{i}.bin"
        piece.write_bytes(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = tmp_path / f"code{i}.o"
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        if co.stat().st_size == 0:
            continue
        dis = subprocess.run([tools[2], "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        for line in dis.splitlines():
            ins = line.split("//")[0].strip()
            if not re.search(r"\bm0\b", ins) or not line.startswith((" ", "\t")):
                continue
            assert re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)", ins), f"unexpected use of M0: {ins!r}"
            n_dma += 1
    assert n_dma > 1000                                      # the unrolled weight-stream pieces of the field kernels


def test_no_packed_fp32_instruction_selects_a_high_half_for_its_low_result(tmp_path):
    """v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with op_sel:[0,1...] (the low result takes the HIGH half of an operand) return a
    wrong low result on the wave's last 16 lanes while another stream's kernel runs v_mfma_f32_32x32x16_{f16,bf16} on the same CUs
    (tools/store_hazard.py: how composite_bwd4_kernel's gradient rows went wrong next to a second refinement
    loop).  hipcc's SLP vectoriser made 98 of them in 53 kernels of this library; it is switched off (csrc/Makefile).  Disassemble the
    library and require that no kernel but the probe that measures the effect carries one."""
    import os, re, subprocess
    from nefes_amd import lib as L
    bindir = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(bindir, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    fat = tmp_path / "fatbin.bin"
    subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", L.LIB_PATH, str(fat)])
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    found, kernels, probe = [], 0, 0
    for i, a in enumerate(starts):
        piece = tmp_path / f"bundle{i}.bin"
        piece.write_bytes(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = tmp_path / f"code{i}.o"
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        if co.stat().st_size == 0:
            continue
        dis = subprocess.run([tools[2], "-d", "-C", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                kernels += 1
                continue
            if name and re.search(r"v_pk_(mul|add|fma)_f32\b.*op_sel:\[0,1", line):
                if "pk_mul_probe_kernel" in name:
                    probe += 1
                else:
                    found.append((name[:90], line.split("//")[0].strip()))
    assert kernels > 100 and probe >= 2, (kernels, probe)          # (the disassembly was read; the probe's two instructions were seen)
    assert not found, found[:5]


def test_no_accumulator_tile_is_relocated_inside_the_asm_scheduled_kernels(tmp_path):
    """The kernels whose MFMAs are asm statements (field_h3.h mma_run_h3_wide: the Wd = 256 fp16 forward instances and the Wd = 256
    inference instances of the backward) rely on the register allocator never moving an accumulator tile while a run is under
    way: an asm statement is instantaneous to hipcc, so a `v_accvgpr_mov` it places behind an issued MFMA copies registers the
    MFMA has not written yet (seen with a seventeenth tile alive; wrong gradients).  Disassemble the
    library, find the asm-scheduled stretches by their entry / exit markers (`s_nop 13` ... `s_nop 12`, used nowhere else) and
    require that none contains an AGPR-to-AGPR move."""
    import os, re, subprocess
    from nefes_amd import lib as L
    bindir = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(bindir, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    fat = tmp_path / "fatbin.bin"
    subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", L.LIB_PATH, str(fat)])
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    checked = {}
    for i, a in enumerate(starts):
        piece = tmp_path / f"bundle{i}.bin"
        piece.write_bytes(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = tmp_path / f"code{i}.o"
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        if co.stat().st_size == 0:
            continue
        dis = subprocess.run([tools[2], "-d", "-C", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                continue
            if name is None:
                continue
            # forward: field_fwd_h3_kernel<MODE, ENC, 256, NTR, false, FH> (inference); backward: field_bwd_h3_kernel<256, KR16, ENC, HAS_T, false, FH>
            wide = (re.match(r"void field_fwd_h3_kernel<\d+, \d+, 256, \d+, false, (true|false)>", name) or
                    re.match(r"void field_bwd_h3_kernel<256, \d+, [012], (true|false), false, (true|false)>", name))     # (HAS_T false: the static-head instances of round 5;
            # ENC 2: the backward with the hash grid in its epilogue -- its first form had a tile moved inside a run, found HERE; round 6: it
            # did it again when the ReLU-mask words changed from shifted to picked, and left the asm schedule: it must carry NO asm run now)
            if not wide:
                continue
            c = checked.setdefault(name, {"mfma": 0, "mov": 0, "runs": 0, "inside": False})
            ins = line.split("//")[0].strip()
            if ins == "s_nop 13":                                   # entry of an asm-scheduled run (field_h3.h mma_run_h3_wide)
                c["inside"] = True
                c["runs"] += 1
            elif ins == "s_nop 12":                                 # its end-of-run statement
                c["inside"] = False
            elif c["inside"]:
                if "v_mfma_f32_32x32x16_f16" in ins:
                    c["mfma"] += 1
                if "v_accvgpr_mov_b32" in ins:
                    c["mov"] += 1
    assert len(checked) >= 5 and any("field_bwd_h3_kernel<256, 2, 2," in n for n in checked), list(checked)                       # sigma / full x two encodings forward, two encodings backward
    for name, c in checked.items():
        if "field_bwd_h3_kernel<256, 2, 2," in name:                # (compiler-placed MFMAs: a moved tile is the compiler's to pad; tests/test_hazard_lint.py)
            assert c["runs"] == 0, (name, c)
            continue
        assert c["runs"] >= 1 and not c["inside"] and c["mfma"] > 900 and c["mov"] == 0, (name, c)


# ---- fp16 two-part streams (layout.h NEFES_STREAM_*_H3, pack.cpp h3 segments, field_h3.h) -------------------------------------
H3F = dict(L1=0, L2=1, L3=2, L4=3, L5H=4, L5E=5, L6=6, L7=7, L8=8, SIG=9, FINAL=10, DT_H=11, DT_D=12, RGB=13, T1=14, T2=15, TH=16)
H3B = dict(RGB=0, TH=1, T2=2, T1=3, T0=4, DIR=5, FINAL=6, SIG=7, L8=8, L7=9, L6=10, L5=11, L4=12, L3=13, L2=14, L1=15)


def _h3_slab_kib(which, Wd=256):
    """KiB per slab of the fp16 two-part streams (layout.h nefes_stream_slab_kib): the Wd=128 forward streams use their own size."""
    txt = open(os.path.join(os.path.dirname(__file__), '..', 'nefes_amd', 'csrc', 'layout.h')).read()
    name = 'NEFES_H3_%s_SLAB_KIB_128' % which if Wd == 128 else 'NEFES_H3_%s_SLAB_KIB' % which
    return int(re.search(r'#define %s (\d+)' % name, txt).group(1))


def pick_exp(m):
    """field_h3.h pick_exp: exponent ex with m 2^ex in [2^14, 2^15), 0 for zero / tiny m (per sample)."""
    m = np.asarray(m, np.float32)
    be = (m.view(np.uint32) >> 23).astype(np.int64)
    return np.where((be < 20) | (be > 250), 0, 14 + 127 - be)


def split_f16(x):
    """(hi, lo) fp16 parts of fp32 values: hi = RNE(x), lo = RNE(x - hi) (v_fma_mixlo/hi_f16 of field_h3.h split_pair_h)."""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


class StreamH3:
    """Consumption of an fp16 two-part stream exactly as mma_run_h3 does: units of two 1 KiB groups (hi | lo), slab_kib/2 units
    per slab, a segment starts on a slab boundary; lane (m, g) of unit (k16-step q, tile t) multiplies slot (8q+i, g).  The
    operand vector arrives in fp32 with a per-sample multiplier 2^ex; the three products hh + hl + lh are accumulated (in
    float64 here; fp32 in the matrix core).  fp32 segments of the same stream (backward heads) go through `mma32`.
    Scale table (layout.h): per segment (weight exponent, row bound), then max |b| per bias block."""

    def __init__(self, blob, si, slab_kib, n_segs):
        self.slab_kib, self.half = slab_kib, slab_kib * 512
        self.ups = slab_kib // 2
        self.raw = np.frombuffer(blob, np.uint16, count=si.n_slabs * self.half, offset=si.slab_off).reshape(si.n_slabs, self.half)
        self.f32 = np.frombuffer(blob, np.float32, count=si.n_slabs * slab_kib * 256, offset=si.slab_off).reshape(si.n_slabs, slab_kib, 64, 4)
        nb = si.scale_off
        self.bias = np.frombuffer(blob, np.float32, count=nb, offset=si.bias_off)
        tab = np.frombuffer(blob, np.uint32, count=si.scale_count, offset=si.bias_off + 4 * nb)
        self.wexp = tab[0:2 * n_segs:2].view(np.int32)
        self.rowb = tab[1:2 * n_segs:2].view(np.float32)
        self.bmax = tab[2 * n_segs:].view(np.float32)
        self.pos = self.bpos = 0

    def bias_tiles(self, nt, n):
        b = self.bias[self.bpos:self.bpos + nt * 32].reshape(nt, 32)
        self.bpos += nt * 32
        return np.repeat(b[:, :, None], n, 2).astype(np.float64).copy()

    def mma(self, nt, vec, ex, acc):
        """acc[t] += (W 2^ew)(vec 2^ex) as hh + hl + lh.  vec [ks, 2, n] fp32 (ks a multiple of 8), ex [n] ints."""
        k16 = vec.shape[0] // 8
        n_units = k16 * nt
        scaled = (vec.astype(np.float64) * np.exp2(ex.astype(np.float64))[None, None, :]).astype(np.float32)
        assert np.all(np.abs(scaled) < 65504.0), "operand exponent lets a value overflow fp16"
        self.headroom = min(getattr(self, "headroom", 99.), float(np.log2(65536.0 / max(np.abs(scaled).max(), 1e-30))))
        bh, bl = split_f16(scaled)
        for u in range(n_units):
            sl, uu = self.pos + u // self.ups, u % self.ups
            q, t = u // nt, u % nt
            unit = self.raw[sl, uu * 1024:(uu + 1) * 1024].reshape(2, 64, 8)
            wh, wl = unit[0].view(np.float16).astype(np.float64), unit[1].view(np.float16).astype(np.float64)   # [lane][i]
            for g in range(2):
                rows = slice(32 * g, 32 * g + 32)
                h_, l_ = bh[8 * q:8 * q + 8, g], bl[8 * q:8 * q + 8, g]
                acc[t] += wl[rows] @ h_ + wh[rows] @ l_ + wh[rows] @ h_
        self.pos += (n_units + self.ups - 1) // self.ups

    def mma32(self, nt, vec, acc):
        ks = vec.shape[0]
        frags = self.slab_kib * 4
        sps = frags // nt
        for sl in range((ks + sps - 1) // sps):
            slab = self.f32[self.pos]
            self.pos += 1
            for f in range(min(sps, ks - sl * sps) * nt):
                s, t = sl * sps + f // nt, f % nt
                a = slab[f // 4, :, f % 4].astype(np.float64)
                acc[t] += a[:32, None] * vec[s, 0][None, :] + a[32:, None] * vec[s, 1][None, :]


def tau_of(M, ew):
    """field_fwd_h3.hip / field_bwd_h3.hip tau_of: exponent the operand is brought to, from the BOUND M of its largest magnitude."""
    return np.minimum(pick_exp(np.asarray(M, np.float32)), 100 - ew)


def relu_max(acc, es):
    """what ReluSplitH measures while splitting: the largest ReLU'd value, back in true units"""
    return (np.maximum(acc, 0).max((0, 1)) * np.exp2(-es.astype(np.float64))).astype(np.float32)


def abs_max(acc, es):
    return (np.abs(acc).max((0, 1)) * np.exp2(-es.astype(np.float64))).astype(np.float32)


@pytest.mark.parametrize("Wd,Cf", [(256, 16), (128, 128), (256, 128), (128, 16), (256, 64), (128, 141)])
def test_h3_streams_reproduce_the_mlp(Wd, Cf):
    """(Wd, C) beyond the two canonical shapes: the head CLASSES of layout.h -- the rgb+feature head padded to 1 or 5 output tiles and
    to 2 or 9 k-steps of its transposed product -- so that C is a run-time parameter of the compiled instances.
    The fp16 two-part streams consumed in kernel order with the kernels' scale bookkeeping (field_fwd_h3.hip /
    field_bwd_h3.hip): operand exponents from BOUNDS of the largest magnitude (the packer's row bounds x the exactly measured
    maximum of the previous operand + max |b|), per-matrix weight exponents from the table, common exponents where two products
    share accumulators, bias x 2^es, outputs x 2^-es.  No operand may overflow fp16 (asserted in StreamH3.mma); forward against
    the float64 oracle to fp32-level accuracy -- sigma-only and full -- and backward-to-inputs against float64 autograd."""
    n = 12
    g = torch.Generator().manual_seed(8)
    pts = (torch.rand(n, 3, generator=g) - .5) * 5
    pts[0] *= 1e-3                                                 # a sample with tiny coordinates
    pts[1] *= 6.                                                   # and one far out (|x| ~ 15)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    e63, e27 = O.freq_encode(pts, 10), O.freq_encode(dirs, 4)
    NTW, NTH, NTR = Wd // 32, Wd // 64, (1 if 3 + Cf <= 32 else 5)          # layout.h nefes_head_ntr
    E, D = emb_vector(e63.numpy(), 10, 32), emb_vector(e27.numpy(), 4, 16)
    mE = np.maximum(1.0, np.abs(pts.numpy()).max(1)).astype(np.float32)
    zeros = np.zeros(n, np.int64)

    def trunk(st, full):
        b = {"L1": st.bias_tiles(NTW, n)}
        for l in range(2, 9):
            b[f"L{l}"] = st.bias_tiles(NTW, n)
        b["SIG"] = st.bias_tiles(1, n)
        if full:
            for name, nt in (("FINAL", NTW), ("DIR", NTH), ("RGB", NTR), ("T0", NTH), ("T1", NTH), ("T2", NTH), ("TH", 1)):
                b[name] = st.bias_tiles(nt, n)
        w, rb, bm = st.wexp, st.rowb, st.bmax
        BB = dict(L1=0, SIG=8, FINAL=9, DIR=10, RGB=11, T0=12, T1=13, T2=14, TH=15)
        masks, out = {}, {}
        tau = tau_of(mE, w[H3F["L1"]])
        es = tau + w[H3F["L1"]]
        acc = b["L1"] * np.exp2(es)[None, None, :]
        st.mma(NTW, E, tau, acc)
        M = rb[H3F["L1"]] * mE + bm[BB["L1"]]

        def sigma(acc, es, M):
            H = acc_to_vec(np.maximum(acc, 0).astype(np.float32))
            tau = tau_of(M, w[H3F["SIG"]])
            e_sg = tau + w[H3F["SIG"]]
            sg = b["SIG"] * np.exp2(e_sg)[None, None, :]
            st.mma(1, H, tau - es, sg)
            out["sigma"] = softplus(sg[0, 0] * np.exp2(-e_sg))

        for l in range(2, 10 if full else 9):
            name = f"L{l}" if l <= 8 else "FINAL"
            seg = H3F["L5H"] if l == 5 else H3F[name]
            assert np.all(relu_max(acc, es) <= M), "row bound violated"
            masks[f"L{l - 1}"] = acc > 0
            if l == 9:
                sigma(acc, es, M)                                   # static_sigma reads the same relu(h8)
            H = acc_to_vec(np.maximum(acc, 0).astype(np.float32))
            mx = relu_max(acc, es)
            tau = tau_of(np.maximum(M, mE) if l == 5 else M, w[seg])
            es_new = tau + w[seg]
            nxt = b[name] * np.exp2(es_new)[None, None, :]
            st.mma(NTW, H, tau - es, nxt)
            M = rb[seg] * mx + bm[BB["FINAL"] if l == 9 else l - 1]
            if l == 5:
                assert w[H3F["L5E"]] == w[H3F["L5H"]]
                st.mma(NTW, E, tau, nxt)
                M = M + rb[H3F["L5E"]] * mE
            acc, es = nxt, es_new
        if not full:
            assert np.all(relu_max(acc, es) <= M)
            sigma(acc, es, M)
            return out, masks
        # acc = xyz_encoding_final output (no activation), exponent es
        assert np.all(abs_max(acc, es) <= M)
        mD = np.abs(D).max((0, 1)).astype(np.float32)
        ew = w[H3F["DT_H"]]
        assert w[H3F["DT_D"]] == ew
        tau = tau_of(np.maximum(M, mD), ew)
        es_dt = tau + ew
        dt = np.concatenate([b["DIR"], b["T0"]], 0) * np.exp2(es_dt)[None, None, :]
        mx = abs_max(acc, es)
        st.mma(2 * NTH, acc_to_vec(acc.astype(np.float32)), tau - es, dt)
        st.mma(2 * NTH, D, tau, dt)
        M = rb[H3F["DT_H"]] * mx + rb[H3F["DT_D"]] * mD + max(bm[BB["DIR"]], bm[BB["T0"]])
        assert np.all(abs_max(dt, es_dt) <= M)
        masks["DIR"], masks["T0"] = dt[:NTH] > 0, dt[NTH:] > 0
        tau = tau_of(M, w[H3F["RGB"]])
        es_ar = tau + w[H3F["RGB"]]
        ar = b["RGB"] * np.exp2(es_ar)[None, None, :]
        st.mma(NTR, acc_to_vec(np.maximum(dt[:NTH], 0).astype(np.float32)), tau - es_dt, ar)
        out["rgbfeat"] = (ar * np.exp2(-es_ar)[None, None, :]).reshape(NTR * 32, n)[:3 + Cf].T
        src, es_s = dt[NTH:], es_dt
        for name in ("T1", "T2"):
            tau = tau_of(M, w[H3F[name]])
            es_n = tau + w[H3F[name]]
            acc2 = b[name] * np.exp2(es_n)[None, None, :]
            mx = relu_max(src, es_s)
            st.mma(NTH, acc_to_vec(np.maximum(src, 0).astype(np.float32)), tau - es_s, acc2)
            M = rb[H3F[name]] * mx + bm[BB[name]]
            assert np.all(relu_max(acc2, es_n) <= M)
            masks[name] = acc2 > 0
            src, es_s = acc2, es_n
        tau = tau_of(M, w[H3F["TH"]])
        es_th = tau + w[H3F["TH"]]
        th = b["TH"] * np.exp2(es_th)[None, None, :]
        st.mma(1, acc_to_vec(np.maximum(src, 0).astype(np.float32)), tau - es_s, th)
        th = th * np.exp2(-es_th)[None, None, :]
        sig = lambda x: 1 / (1 + np.exp(-x))
        out["t_rgb"], out["t_sigma"], out["t_beta"] = sig(th[0, :3]).T, softplus(th[0, 3]), softplus(th[0, 4])
        assert st.pos == st.raw.shape[0] and st.bpos == st.bias.shape[0]
        return out, masks

    # sigma-only stream of the coarse net
    pc, info_c, blob_c = pack(Wd, Cf, "coarse")
    si = info_c.stream[L.STREAM_FWD_SIGMA_H3]
    assert si.n_slabs > 0 and si.scale_count >= 2 * 10 + 9
    st = StreamH3(blob_c, si, _h3_slab_kib("FWD", Wd), 10)
    out, _ = trunk(st, False)
    p64 = {k: v.double() for k, v in pc.items()}
    ref = O.field_forward(p64, e63.double(), sigma_only=True)[:, 0].numpy()
    assert np.abs(out["sigma"] - ref).max() <= 2e-6 * np.abs(ref).max()

    # full stream of the fine net
    pf, info_f, blob_f = pack(Wd, Cf, "fine")
    st = StreamH3(blob_f, info_f.stream[L.STREAM_FWD_FULL_H3], _h3_slab_kib("FWD", Wd), 17)
    out, masks = trunk(st, True)
    print(f"least fp16 headroom of any operand (binades below 2^16): {st.headroom:.1f}")
    p64 = {k: v.double() for k, v in pf.items()}
    emb = torch.cat([e63, e27], 1).double().requires_grad_()
    raw = O.field_forward(p64, emb, output_transient=True)
    r = raw.detach().numpy()
    C3 = 3 + Cf
    got = np.concatenate([out["rgbfeat"], out["sigma"][:, None], out["t_rgb"], out["t_sigma"][:, None], out["t_beta"][:, None]], 1)
    err = np.abs(got - r).max(0) / np.abs(r).max(0)
    assert err.max() <= 3e-6, err                                   # fp32-level (the fp32 oracle itself: ~1e-6)

    # backward-to-inputs on the fp16 stream, against float64 autograd
    g_raw = torch.randn(raw.shape, generator=g).double()
    (g_emb,) = torch.autograd.grad(raw, emb, g_raw)
    gr = g_raw.numpy()
    d_pre = {"rgbfeat": gr[:, :C3], "sigma": gr[:, C3] * (1 - np.exp(-r[:, C3])),
             "t_rgb": gr[:, C3 + 1:C3 + 4] * r[:, C3 + 1:C3 + 4] * (1 - r[:, C3 + 1:C3 + 4]),
             "t_sigma": gr[:, C3 + 4] * (1 - np.exp(-r[:, C3 + 4])), "t_beta": gr[:, C3 + 5] * (1 - np.exp(-r[:, C3 + 5]))}
    st = StreamH3(blob_f, info_f.stream[L.STREAM_BWD_FULL_H3], _h3_slab_kib("BWD", Wd), 16)
    w, rb = st.wexp, st.rowb
    assert st.bias.size == 0 and w[H3B["T0"]] == w[H3B["DIR"]] and all(w[H3B[k]] == 0 for k in ("TH", "SIG"))
    Z = lambda nt: np.zeros((nt, 32, n), np.float64)
    f32v = lambda v: v.astype(np.float32)
    G2 = Z(NTH)
    # static_rgb^T: an fp16 product too (round 3): the head class's k-steps of 16 (layout.h nefes_head_kr16), natural slots, exponent
    # from the operand's exact maximum
    KR16 = 2 if C3 <= 32 else 9
    dr = np.zeros((8 * KR16, 2, n), np.float32)
    for e in range(8 * KR16):
        for h in range(2):
            ch = 32 * (e >> 4) + rho(h, e & 15)
            if ch < C3:
                dr[e, h] = d_pre["rgbfeat"][:, ch]
    M_dr = np.abs(d_pre["rgbfeat"]).max(1).astype(np.float32)
    tau = tau_of(M_dr, w[H3B["RGB"]])
    es_g2 = tau + w[H3B["RGB"]]
    st.mma(NTH, dr, tau, G2)
    M_g2 = rb[H3B["RGB"]] * M_dr
    assert np.all(abs_max(G2, es_g2) <= M_g2)
    T3 = Z(NTH)
    dth = [d_pre["t_rgb"][:, 0], d_pre["t_rgb"][:, 1], d_pre["t_rgb"][:, 2], d_pre["t_sigma"], d_pre["t_beta"]]
    st.mma32(NTH, compact(dth, 3), T3)
    M = rb[H3B["TH"]] * np.abs(np.stack(dth, 1)).max(1).astype(np.float32)
    src, es = T3, zeros
    for name, mk in (("T2", "T2"), ("T1", "T1")):                   # transient_encoding.4^T, .2^T
        assert np.all(abs_max(src, es) <= M)
        tau = tau_of(M, w[H3B[name]])
        dst = Z(NTH)
        mx = abs_max(src * masks[mk], es)
        st.mma(NTH, acc_to_vec(f32v(src * masks[mk])), tau - es, dst)
        M = rb[H3B[name]] * mx
        src, es = dst, tau + w[H3B[name]]
    tau = tau_of(np.maximum(M, M_g2), w[H3B["T0"]])
    es_dt = tau + w[H3B["T0"]]
    a9 = Z(NTW + 1)
    mt, mg = abs_max(src * masks["T0"], es), abs_max(G2 * masks["DIR"], es_g2)
    st.mma(NTW + 1, acc_to_vec(f32v(src * masks["T0"])), tau - es, a9)
    st.mma(NTW + 1, acc_to_vec(f32v(G2 * masks["DIR"])), tau - es_g2, a9)
    M = rb[H3B["T0"]] * mt + rb[H3B["DIR"]] * mg
    assert np.all(abs_max(a9, es_dt) <= M)
    dD = acc_to_vec(f32v(a9), 0, 1) * np.exp2(-es_dt)[None, None, :]
    tau = tau_of(M, w[H3B["FINAL"]])
    es = tau + w[H3B["FINAL"]]
    acc = Z(NTW)
    mx = abs_max(a9[1:], es_dt)
    st.mma(NTW, acc_to_vec(f32v(a9), 1, NTW), tau - es_dt, acc)
    st.mma32(NTW, compact([d_pre["sigma"] * np.exp2(es)], 1), acc)
    M = rb[H3B["FINAL"]] * mx + rb[H3B["SIG"]] * np.abs(d_pre["sigma"]).astype(np.float32)
    accE, es_e = None, None
    for l in range(8, 1, -1):
        assert np.all(abs_max(acc, es) <= M)
        tau = tau_of(M, w[H3B[f"L{l}"]])
        masked = acc * masks[f"L{l}"]
        mx = abs_max(masked, es)
        Hm = acc_to_vec(f32v(masked))
        es_new = tau + w[H3B[f"L{l}"]]
        if l == 5:
            a10 = Z(NTW + 2)
            st.mma(NTW + 2, Hm, tau - es, a10)
            accE, es_e, acc = a10[:2].copy(), es_new, a10[2:]
        else:
            acc = Z(NTW)
            st.mma(NTW, Hm, tau - es, acc)
        M = rb[H3B[f"L{l}"]] * mx
        es = es_new
    tau = tau_of(M, w[H3B["L1"]])
    es1 = tau + w[H3B["L1"]]
    accE = accE * np.exp2(es1 - es_e)[None, None, :]
    st.mma(2, acc_to_vec(f32v(acc * masks["L1"])), tau - es, accE)
    assert st.pos == st.raw.shape[0]
    print(f"backward: least fp16 headroom of any operand: {st.headroom:.1f} binades")
    g63 = emb_vector_T(acc_to_vec(f32v(accE * np.exp2(-es1)[None, None, :])), 10, 63)
    g27 = emb_vector_T(f32v(dD[:14]), 4, 27)
    scale = np.abs(g_emb.numpy()).max(1, keepdims=True)
    assert (np.abs(g63 - g_emb.numpy()[:, :63]) / scale).max() <= 5e-6
    assert (np.abs(g27 - g_emb.numpy()[:, 63:]) / scale).max() <= 5e-6


def test_h3_stream_decodes_to_scaled_weights():
    """Every weight of an fp16 segment is stored as (hi, lo) = (RNE_f16(w 2^e), RNE_f16(w 2^e - hi)) with e from the stream's
    exponent table and max |w| 2^e in [2^14, 2^15): hi + lo reproduces w to 2^-22 of the matrix' largest weight."""
    Wd, Cf = 256, 16
    p, info, blob = pack(Wd, Cf, "coarse")
    si = info.stream[L.STREAM_FWD_SIGMA_H3]
    st = StreamH3(blob, si, _h3_slab_kib("FWD", Wd), 10)
    nt, k16, ups = Wd // 32, Wd // 16, st.ups
    l1_slabs = (4 * nt + ups - 1) // ups
    Wm = p["xyz_encoding_2.0.weight"].numpy().astype(np.float64)
    e = int(st.wexp[H3F["L2"]])
    assert 2.0 ** 14 <= np.abs(Wm).max() * 2.0 ** e < 2.0 ** 15
    got = np.zeros_like(Wm)
    for u in range(k16 * nt):
        sl, uu = l1_slabs + u // ups, u % ups
        q, t = u // nt, u % nt
        unit = st.raw[sl, uu * 1024:(uu + 1) * 1024].reshape(2, 64, 8)
        val = unit[0].view(np.float16).astype(np.float64) + unit[1].view(np.float16).astype(np.float64)
        for lane in range(64):
            m_, g_ = lane & 31, lane >> 5
            for i in range(8):
                s = 8 * q + i
                got[32 * t + m_, 32 * (s >> 4) + rho(g_, s & 15)] = val[lane, i]
    assert np.abs(got * 2.0 ** -e - Wm).max() <= 2.0 ** -22 * np.abs(Wm).max()
