"""Train mode of the field MLP: forward with saved pre-activations + weight gradients.

What `loss.backward()` does in script/run_nefes.py:42-108 for the NeRF weights, on the HIP path:

    forward   nefes_field_fwd_train   the fused forward kernel (TRAIN instance) also writes every hidden layer's
                                      pre-activation and both embeddings to `acts` [tiles][rows][128]
    backward  nefes_train_head_grad   d raw -> head pre-activation gradients
              nefes_train_dx          G_{l-1} = relu'(pre_{l-1}) * W_l^T G_l, layer by layer (MFMA)
              nefes_train_dw          dW_l = G_l act(X_{l-1})^T over all samples (MFMA, split over sample tiles)
              bias gradients          row sums of `dacts`

The layer graph below is the reference's NeRFH_NFF.forward (models/nerfh_nff.py:525-576): trunk 1..8 with the
skip `cat([input_xyz, h])` at layer 5, static_sigma on h8, xyz_encoding_final (no activation),
dir_encoding / transient_encoding.0 on `cat([final, dir_emb])`, static_rgb, transient_encoding.{2,4}, transient heads.
Gradients w.r.t. the rays are not produced here (training rays are data; the pose-refinement path uses field_bwd).
"""
import ctypes as C

import torch

from . import lib as L
from . import ops

EMB_XYZ, EMB_DIR = 63, 27
DEBUG = None          # tests may set this to a dict to receive the acts / dacts buffers of the last backward pass


def rows_view(buf):
    """[tiles, rows, 128] copy of an acts / dacts buffer in (row, sample) order.  The device layout stores, inside a tile, blocks
    of 32 rows x 16 samples contiguously (csrc/layout.h nefes_train_off): [row / 32][sample / 16][row % 32][sample % 16]."""
    T, rows, _ = buf.shape
    return buf.reshape(T, rows // 32, 8, 32, 16).permute(0, 1, 3, 2, 4).reshape(T, rows, 128)


def _emb_slot(n_freq, s, h):
    """layout.h nefes_emb_slot: embedding slot (s, h) -> index in the reference's embedding order, -1 = padding."""
    if s < 3 * n_freq:
        return 3 + 6 * (s // 3) + 3 * h + (s % 3)
    if s == 3 * n_freq:
        return h
    if s == 3 * n_freq + 1:
        return 2 if h == 0 else -1
    return -1


def _slot_rows(n_freq, n_rows, n_nat, device):
    """index tensor: natural embedding feature -> row of the slot-ordered block."""
    idx = [-1] * n_nat
    for row in range(n_rows):
        k = _emb_slot(n_freq, row // 2, row % 2)
        if k >= 0:
            idx[k] = row
    assert min(idx) >= 0
    return torch.tensor(idx, device=device)


def _pad_t(w, n_in_pad, k_pad):
    """[out, in] weight -> transposed, zero-padded [n_in_pad, k_pad] contiguous (A operand of nefes_train_dx)."""
    out, inn = w.shape
    wt = torch.zeros(n_in_pad, k_pad, device=w.device, dtype=torch.float32)
    wt[:inn, :out] = w.detach().t()
    return wt


def _dw_grid(ot, it):
    """(workgroups per share of the sample tiles, waves wanted per launch) for an ot x it-tile weight gradient: mirrors the
    block shapes nefes_train_dw picks (csrc/train.hip).  The big blocks hold 128-256 accumulator registers: one wave per SIMD,
    1024 SIMDs; the small ones run two or three waves per SIMD."""
    if ot % 4 == 0 and it % 4 == 0:
        return (ot // 4) * (it // 4), 1024
    if ot == 5 and it == 2:
        return 1, 1024
    if ot % 4 == 0 and it == 2:
        return ot // 4, 1024
    nto = 2 if ot % 2 == 0 else 1
    nti = 4 if it % 4 == 0 else (2 if it % 2 == 0 else 1)
    return (ot // nto) * (it // nti), 2048


class _Pass:
    def __init__(self, net, desc, n_tiles, acts, dacts):
        self.lib, self.desc, self.n_tiles = L.load(), desc, n_tiles
        self.acts, self.dacts = acts, dacts
        self.rows = int(self.lib.nefes_train_rows(C.byref(desc)))
        self.offsets = [int(self.lib.nefes_train_row_offset(C.byref(desc), b)) for b in range(L.TB_END + 1)]
        self.stream = ops._stream()
        self.dev = acts.device
        self.db = {}                                                  # block -> bias gradient (filled by run)
        self.jobs, self.cols = [], 0                                  # declared weight-gradient products, columns of the wide partial buffer

    def off(self, block):                                             # (a method, not a closure: no reference cycle
        return self.offsets[block]                                    #  may keep the multi-GB buffers alive)

    def dx(self, g_block, n_out, w, n_in, dst_block, accumulate, mask, g_rows=None):
        k = (n_out + 7) // 8 * 8
        wt = _pad_t(w, n_in, k)
        L.check(self.lib.nefes_train_dx(self.n_tiles, self.rows, self.dacts.data_ptr(), self.off(g_block), k, wt.data_ptr(), k,
                                        n_in, self.acts.data_ptr(), self.off(dst_block), int(accumulate), int(mask),
                                        self.dacts.data_ptr(), self.stream), "nefes_train_dx")
        return wt                                                     # kept alive by the caller until the stream is done

    def dw(self, g_block, n_out, x_block, n_in, relu, bias=True):
        """Declares sum_s G[o][s] f(X[i][s]) -> handle; handle.v = [n_out_pad, n_in_pad] after run().  The row sums of G (the
        Linear's bias gradient) ride along as one more column of every product's partials; bias=True keeps them in
        self.db[g_block]."""
        op, ip = (n_out + 31) // 32 * 32, (n_in + 31) // 32 * 32
        h = _Product(g_block, op, x_block, ip, bool(relu), bool(bias), self.cols)
        self.cols += op * (ip + 1)
        self.jobs.append(h)
        return h

    def run(self):
        """Launches every declared product into ONE wide partial buffer [splits][all products' columns] (same number of shares
        for all, nefes_train_dw_bias' split_stride) and sums the shares with a single reduction, in a fixed order."""
        big = [_dw_grid(h.op // 32, h.ip // 32)[0] for h in self.jobs if _dw_grid(h.op // 32, h.ip // 32)[1] == 1024]
        splits = max(1, min(self.n_tiles, 1024 // max(big + [1])))
        wide = torch.empty(splits, self.cols, device=self.dev)
        for h in self.jobs:
            L.check(self.lib.nefes_train_dw_bias(self.n_tiles, self.rows, self.dacts.data_ptr(), self.off(h.g_block), h.op,
                                                 self.acts.data_ptr(), self.off(h.x_block), h.ip, int(h.relu), splits, self.cols,
                                                 wide.data_ptr() + 4 * h.col, self.stream), "nefes_train_dw_bias")
        red = wide.sum(0)
        for h in self.jobs:
            out = red[h.col:h.col + h.op * (h.ip + 1)].view(h.op, h.ip + 1)
            h.v = out[:, :h.ip]
            if h.bias:
                self.db[h.g_block] = out[:, h.ip]
        self.jobs = []


class _Product:
    __slots__ = ("g_block", "op", "x_block", "ip", "relu", "bias", "col", "v")

    def __init__(self, g_block, op, x_block, ip, relu, bias, col):
        self.g_block, self.op, self.x_block, self.ip, self.relu, self.bias, self.col, self.v = g_block, op, x_block, ip, relu, bias, col, None


def param_names(net, mode):
    names = [f"xyz_encoding_{i}.0" for i in range(1, 9)] + ["xyz_encoding_final", "dir_encoding.0", "static_sigma.0",
                                                             "static_rgb.0"]
    if mode == L.FIELD_FULL:
        names += ["transient_encoding.0", "transient_encoding.2", "transient_encoding.4", "transient_sigma.0",
                  "transient_rgb.0", "transient_beta.0"]
    return [n + s for n in names for s in (".weight", ".bias")]


def fp16_pipe(pk):
    """The train-mode forward (and the fused dX chain) run on the fp16 two-part instances when ops.SPLIT selects them, the
    network's fp16 streams are current, and the shape has them (both widths x both head classes, frequency embedding)."""
    return ops._h3(pk) and ops.h3_shape(pk) and pk.xyz_encoding == L.XYZ_FREQ10


FUSED_DX = True       # one fused backward launch (nefes_field_bwd_train) instead of the layer-by-layer nefes_train_dx chain


def weight_grads(net, pk, mode, N, S, raw_t, g_raw_t, acts, fused=None):
    """-> dict parameter name -> gradient (fp32, parameter shape) for every parameter on the path of `mode`.
    fused = (rays_o, rays_d, viewdirs, z, masks): run the dX chain as one fused kernel launch."""
    lib, desc = L.load(), pk.desc
    W, Cf = net.W, net.W_features
    C3, H2 = 3 + Cf, W // 2
    full = mode == L.FIELD_FULL
    n_tiles = acts.shape[0]
    dacts = torch.empty_like(acts)
    if not (fused is not None and fp16_pipe(pk)):             # (the fp16 dX kernel writes the head blocks itself)
        L.check(lib.nefes_train_head_grad(C.byref(desc), mode, N, S, raw_t.data_ptr(), g_raw_t.data_ptr(), dacts.data_ptr(),
                                          ops._stream()), "nefes_train_head_grad")
    P = _Pass(net, desc, n_tiles, acts, dacts)
    sd = dict(net.named_parameters())
    w = lambda name: sd[name + ".weight"]
    keep = []
    TB = lambda l: L.TB_L1 + (l - 1)
    # ---- backward through the layers: dacts rows of every hidden block become d loss / d pre-activation ----
    if fused is not None:
        o, d, v, zz, masks = fused
        g_pts, g_vs = torch.empty(N * S, 3, device=acts.device), torch.empty(N * S, 3, device=acts.device)
        bwd = lib.nefes_field_bwd_train_h3 if fp16_pipe(pk) else lib.nefes_field_bwd_train
        L.check(bwd(C.byref(desc), pk.blob.data_ptr(), mode, N, S, o.data_ptr(), d.data_ptr(), zz.data_ptr(), v.data_ptr(),
                    raw_t.data_ptr(), g_raw_t.data_ptr(), masks.data_ptr(), dacts.data_ptr(), g_pts.data_ptr(), g_vs.data_ptr(),
                    ops._stream()), "nefes_field_bwd_train")
    else:
        if net._embedding_columns():
            # the layer-by-layer chain slices the network's OWN weights at the paper-default embedding widths (63 / 27); a network on
            # fewer octaves has narrower layer-5 / dir_encoding inputs and would hand P.dx the wrong block
            raise NotImplementedError("nefes_amd: networks on a reduced embedding (--reduce_embedding / smaller --multires) train on the "
                                      "fused dX chain only (train.FUSED_DX = True, frequency embedding)")
        if full:
            w_th = torch.cat([w("transient_rgb.0"), w("transient_sigma.0"), w("transient_beta.0")], 0)     # raw channel order
            keep.append(P.dx(L.TB_TH, 5, w_th, H2, L.TB_T2, False, True))
            keep.append(P.dx(L.TB_T2, H2, w("transient_encoding.4"), H2, L.TB_T1, False, True))
            keep.append(P.dx(L.TB_T1, H2, w("transient_encoding.2"), H2, L.TB_T0, False, True))
        keep.append(P.dx(L.TB_RGB, C3, w("static_rgb.0"), H2, L.TB_DIR, False, True))
        keep.append(P.dx(L.TB_DIR, H2, w("dir_encoding.0")[:, :W], W, L.TB_FINAL, False, False))
        if full:
            keep.append(P.dx(L.TB_T0, H2, w("transient_encoding.0")[:, :W], W, L.TB_FINAL, True, False))
        keep.append(P.dx(L.TB_FINAL, W, w("xyz_encoding_final"), W, TB(8), False, False))
        keep.append(P.dx(L.TB_SIG, 1, w("static_sigma.0"), W, TB(8), True, True))
        for l in range(8, 1, -1):
            wl = w(f"xyz_encoding_{l}.0")
            keep.append(P.dx(TB(l), W, wl[:, EMB_XYZ:] if l == 5 else wl, W, TB(l - 1), False, True))
    # ---- weight gradients ----
    e_idx = _slot_rows(10, 64, EMB_XYZ, acts.device)
    d_idx = _slot_rows(4, 28, EMB_DIR, acts.device)
    g = {}
    h1 = P.dw(TB(1), W, L.TB_E, 64, False)
    hl = {l: P.dw(TB(l), W, TB(l - 1), W, True) for l in range(2, 9)}
    h5e = P.dw(TB(5), W, L.TB_E, 64, False, bias=False)
    hsig, hfin = P.dw(L.TB_SIG, 1, TB(8), W, True), P.dw(L.TB_FINAL, W, TB(8), W, True)
    hdir, hdird = P.dw(L.TB_DIR, H2, L.TB_FINAL, W, False), P.dw(L.TB_DIR, H2, L.TB_DV, 32, False, bias=False)
    hrgb = P.dw(L.TB_RGB, C3, L.TB_DIR, H2, True)
    if full:
        ht0, ht0d = P.dw(L.TB_T0, H2, L.TB_FINAL, W, False), P.dw(L.TB_T0, H2, L.TB_DV, 32, False, bias=False)
        ht1, ht2 = P.dw(L.TB_T1, H2, L.TB_T0, H2, True), P.dw(L.TB_T2, H2, L.TB_T1, H2, True)
        hth = P.dw(L.TB_TH, 5, L.TB_T2, H2, True)
    P.run()
    g["xyz_encoding_1.0.weight"] = h1.v[:W][:, e_idx]
    for l in range(2, 9):
        dh = hl[l].v[:W, :W]
        if l == 5:
            dh = torch.cat([h5e.v[:W][:, e_idx], dh], 1)
        g[f"xyz_encoding_{l}.0.weight"] = dh
    g["static_sigma.0.weight"] = hsig.v[:1, :W]
    g["xyz_encoding_final.weight"] = hfin.v[:W, :W]
    g["dir_encoding.0.weight"] = torch.cat([hdir.v[:H2, :W], hdird.v[:H2][:, d_idx]], 1)
    g["static_rgb.0.weight"] = hrgb.v[:C3, :H2]
    if full:
        g["transient_encoding.0.weight"] = torch.cat([ht0.v[:H2, :W], ht0d.v[:H2][:, d_idx]], 1)
        g["transient_encoding.2.weight"] = ht1.v[:H2, :H2]
        g["transient_encoding.4.weight"] = ht2.v[:H2, :H2]
        d_th = hth.v[:5, :H2]
        g["transient_rgb.0.weight"], g["transient_sigma.0.weight"], g["transient_beta.0.weight"] = d_th[:3], d_th[3:4], d_th[4:5]
    # ---- bias gradients: row sums of the gradient blocks over all samples, summed inside the dW launches ----
    blk = lambda b, n: P.db[b][:n]
    for l in range(1, 9):
        g[f"xyz_encoding_{l}.0.bias"] = blk(TB(l), W)
    g["xyz_encoding_final.bias"], g["dir_encoding.0.bias"] = blk(L.TB_FINAL, W), blk(L.TB_DIR, H2)
    g["static_sigma.0.bias"], g["static_rgb.0.bias"] = blk(L.TB_SIG, 1), blk(L.TB_RGB, C3)
    if full:
        g["transient_encoding.0.bias"], g["transient_encoding.2.bias"] = blk(L.TB_T0, H2), blk(L.TB_T1, H2)
        g["transient_encoding.4.bias"] = blk(L.TB_T2, H2)
        th = blk(L.TB_TH, 5)
        g["transient_rgb.0.bias"], g["transient_sigma.0.bias"], g["transient_beta.0.bias"] = th[:3], th[3:4], th[4:5]
    if DEBUG is not None:
        DEBUG.update(acts=rows_view(acts), dacts=rows_view(dacts), rows=P.rows, off={b: P.off(b) for b in range(19)})
    del keep
    if fused is not None:
        g["__rays__"] = (g_pts, g_vs)          # per-sample d pts / d viewdirs of the fused dX chain (joint pose + weight gradients)
    return g


class FieldTrain(torch.autograd.Function):
    """raw_t [N,R,S] of the field at samples z along the rays, differentiable w.r.t. the network parameters
    (passed as *params in `param_names` order so autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, z, net, mode, *params):
        pk = net.packed()
        lib = L.load()
        if mode not in (L.FIELD_STATIC, L.FIELD_FULL):
            raise ValueError("nefes_amd: train mode evaluates the static or the full head")
        o, d, v, zz = ops._f32(rays_o), ops._f32(rays_d), ops._f32(viewdirs), ops._f32(z)
        N, S = zz.shape
        R = 3 + pk.feat_dim + (1 if mode == L.FIELD_STATIC else 6)
        n_tiles = (N * S + 127) // 128
        rows = int(lib.nefes_train_rows(C.byref(pk.desc)))
        raw_t = torch.empty(N, R, S, device=zz.device)
        acts = torch.empty(n_tiles, rows, 128, device=zz.device)
        fused = FUSED_DX and pk.xyz_encoding == L.XYZ_FREQ10
        masks = torch.empty(pk.mask_bytes(N * S) // 4, dtype=torch.int32, device=zz.device) if fused else None
        h3 = fused and fp16_pipe(pk)
        if not h3:
            ops.require_instance(pk, "the fp32-MFMA train-mode forward")
            if not ops.canonical_shape(pk):
                raise RuntimeError(f"nefes_amd: train mode for W={pk.width}, f_dim={pk.feat_dim} runs on the fused fp16 two-part "
                                   f"instances only (train.FUSED_DX, NEFES_SPLIT=h3).  Compiled: {ops.COMPILED_SET}")
        fwd = lib.nefes_field_fwd_train_h3 if h3 else lib.nefes_field_fwd_train
        with ops._timed("field_fwd_train[h3]" if h3 else "field_fwd_train"):
            L.check(fwd(C.byref(pk.desc), pk.blob.data_ptr(), mode, N, S, ops._chk(o, "rays_o"), ops._chk(d, "rays_d"),
                        ops._chk(zz, "z"), None, ops._chk(v, "viewdirs"), raw_t.data_ptr(), acts.data_ptr(),
                        None if masks is None else masks.data_ptr(), ops._stream()), "nefes_field_fwd_train")
        if fused:
            ops._tap("masks", (masks, N, S, pk.width, mode))
            ctx.save_for_backward(raw_t, acts, o, d, v, zz, masks)
        else:
            ctx.save_for_backward(raw_t, acts)
        ctx.fused = fused
        if DEBUG is not None:
            DEBUG.update(acts=rows_view(acts), rows=rows, off={b: int(lib.nefes_train_row_offset(C.byref(pk.desc), b)) for b in range(19)})
        ctx.net, ctx.pk, ctx.mode, ctx.NS, ctx.pk_gen = net, pk, mode, (N, S), pk.generation
        return raw_t

    @staticmethod
    def backward(ctx, g_raw_t):
        want_rays = any(ctx.needs_input_grad[:3])
        if ctx.needs_input_grad[3] or (want_rays and not ctx.fused):
            raise NotImplementedError("nefes_amd: gradients w.r.t. the rays in train mode come from the fused dX chain "
                                      "(train.FUSED_DX, frequency embedding); the depths z carry no gradient (rendering.py:139 detaches them)")
        ctx.pk.check_generation(ctx.pk_gen)
        raw_t, acts = ctx.saved_tensors[:2]
        N, S = ctx.NS
        with ops._timed("field_bwd_train[h3]" if (ctx.fused and fp16_pipe(ctx.pk)) else "field_bwd_train"):
            g = weight_grads(ctx.net, ctx.pk, ctx.mode, N, S, raw_t, ops._f32(g_raw_t), acts,
                             fused=tuple(ctx.saved_tensors[2:]) if ctx.fused else None)
        names = param_names(ctx.net, ctx.mode)
        g = ctx.net.shrink_grads(g)            # a network on fewer embedding octaves: drop the columns the kernels padded (field.py)
        g_rays = (None, None, None)
        if want_rays:
            # joint pose + weight gradients (e.g. BARF-style training): the fused dX chain wrote d pts / d viewdirs per sample anyway
            zz = ctx.saved_tensors[5]
            g_pts, g_vs = g["__rays__"]
            g_rays = ops.ray_grad_reduce(N, S, zz, g_pts, g_vs)
        return g_rays + (None,) * 3 + tuple(g[n].contiguous() for n in names)


def field_train(net, mode, rays_o, rays_d, viewdirs, z):
    sd = dict(net.named_parameters())
    return FieldTrain.apply(rays_o, rays_d, viewdirs, z, net, mode, *[sd[n] for n in param_names(net, mode)])
