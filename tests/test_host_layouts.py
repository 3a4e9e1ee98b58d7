"""Host-side layout logic that needs no GPU: the train-buffer row view, the packed convolution weights and the identity the
input-gradient launch of csrc/conv.hip relies on."""
import numpy as np
import torch


def test_rows_view_inverts_the_blocked_tile_layout():
    """csrc/layout.h nefes_train_off: element (row, sample) of a tile at [row / 32][sample / 16][row % 32][sample % 16]."""
    from nefes_amd.train import rows_view
    T, rows = 3, 96
    buf = torch.arange(T * rows * 128, dtype=torch.float32).reshape(T, rows, 128)     # value = flat offset inside the buffer
    v = rows_view(buf)
    r, s = np.meshgrid(np.arange(rows), np.arange(128), indexing="ij")
    off = ((r >> 5) * 8 + (s >> 4)) * 512 + (r & 31) * 16 + (s & 15)
    for t in range(T):
        assert np.array_equal(v[t].numpy().astype(np.int64), t * rows * 128 + off)


def test_packed_conv_weights_and_the_gradient_identity():
    """ops._pack_conv: W[co][ci][ty][tx] at [ci][ty*K + tx][co], zero padded; backward = the flipped, transposed weights, with
    which a plain 'same' convolution of the output gradient IS the input gradient (what FrozenConv2d.backward launches)."""
    from nefes_amd import ops
    g = torch.Generator().manual_seed(0)
    for cout, cin, k in ((5, 3, 3), (7, 4, 5)):
        w = torch.randn(cout, cin, k, k, generator=g)
        pk = ops._pack_conv(w, False)
        assert pk.shape == ((cin + 1) // 2 * 2, k * k, 32)
        for co, ci, ty, tx in ((0, 0, 0, 0), (cout - 1, cin - 1, k - 1, 0), (2, 1, 1, k - 1)):
            assert float(pk[ci, ty * k + tx, co]) == float(w[co, ci, ty, tx])
        assert float(pk[:, :, cout:].abs().max()) == 0.0 and float(pk[cin:].abs().sum()) == 0.0
        pb = ops._pack_conv(w, True)
        assert pb.shape == ((cout + 1) // 2 * 2, k * k, 32)
        assert float(pb[1, 0, 2]) == float(w[1, 2, k - 1, k - 1])                      # W'[ci=2][co=1][0][0] = W[1][2][K-1][K-1]
        x = torch.randn(2, cin, 6, 7, generator=g, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(x, w.double(), padding=k // 2)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        (y * gy).sum().backward()
        wb = w.double().flip(2, 3).transpose(0, 1)
        assert torch.allclose(torch.nn.functional.conv2d(gy, wb, padding=k // 2), x.grad, atol=1e-12)
        assert ops._pack_conv(w, False) is pk                                            # cached per tensor and version
        w.mul_(2.0)
        assert ops._pack_conv(w, False) is not pk                                        # a new version re-packs


def test_dw_grid_covers_every_product_shape():
    """train._dw_grid: blocks per share and waves wanted for every (out tiles, in tiles) the weight-gradient pass launches."""
    from nefes_amd.train import _dw_grid
    for W, C in ((128, 128), (256, 16)):
        H2, C3 = W // 2, 3 + C
        shapes = [(W, 64), (W, W), (1, W), (H2, W), (H2, 32), (C3, H2), (H2, H2), (5, H2)]
        for n_out, n_in in shapes:
            ot, it = (n_out + 31) // 32, (n_in + 31) // 32
            blocks, want = _dw_grid(ot, it)
            assert blocks >= 1 and want in (1024, 2048) and blocks <= ot * it


def test_factored_head_applies_only_where_its_algebra_pays_and_holds():
    """NeRFH_NFF.factored_head_ok (round 5, csrc/field_fwd_h3.hip FH): a FROZEN fine network of width 128 on the frequency embedding whose
    rgb + feature head has more outputs than relu(dir_encoding) has features -- the reference's default (nerfh_nff.py:21,427: 128 feature
    channels on 64) -- and nothing else: trainable weights need the per-sample head for their gradient, the other shapes gain nothing."""
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    fine = lambda Wd, C, **k: NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True, **k)
    assert fine(128, 128).requires_grad_(False).factored_head_ok()
    assert fine(128, 96).requires_grad_(False).factored_head_ok()
    assert not fine(128, 128).factored_head_ok()                                            # trainable
    assert not fine(128, 16).requires_grad_(False).factored_head_ok()                       # 19 outputs against 64 + 1 + 3
    assert not fine(256, 16).requires_grad_(False).factored_head_ok()                       # the headline network
    assert not fine(256, 128).requires_grad_(False).factored_head_ok()                      # 131 against 129: nothing to gain
    assert not fine(128, 128, in_channels_xyz=32).requires_grad_(False).factored_head_ok()  # hash-grid input
    assert not NeRFH_NFF('coarse', W=128, f_dim=128).requires_grad_(False).factored_head_ok()   # no transient head: the static-head instances
    net = fine(128, 128).requires_grad_(False)
    for switch, attr in (("FACTORED_HEAD", False), ("SPLIT", "x6")):
        old = getattr(ops, switch)
        setattr(ops, switch, attr)
        try:
            assert not net.factored_head_ok()
        finally:
            setattr(ops, switch, old)
    # only the FIELD's parameters count as "trainable": the fusion CNN and the exposure network hang off the same module
    for n, p in net.named_parameters():
        p.requires_grad_(n.startswith(("fusion_net", "exposure_embedding")))
    assert net.factored_head_ok()


def test_feature_head_composed_into_fusion_nets_first_convolution_is_exact_algebra():
    """FusionNet._conv0_on_gmap (nefes_amd/field.py; the refinement loop's path): Conv2d(3 + C, 64, 3) applied to
    (rgb, W_f g + b_f a) equals the composed weights applied to (rgb, g, a) -- zero padding included -- in float64 on the CPU."""
    import torch
    from nefes_amd.field import FusionNet
    torch.manual_seed(3)
    C, Cg, H, W = 12, 5, 7, 9
    fnet = FusionNet(C).double()
    w_f, b_f = torch.randn(C, Cg, dtype=torch.float64), torch.randn(C, dtype=torch.float64)
    rgb, g, a = (torch.randn(2, n, H, W, dtype=torch.float64) for n in (3, Cg, 1))
    feat = torch.einsum("cj,bjyx->bcyx", w_f, g) + b_f[None, :, None, None] * a
    c0 = fnet.net[0]
    ref = torch.nn.functional.conv2d(torch.cat([rgb, feat], 1), c0.weight, c0.bias, padding=1)
    w = fnet._conv0_on_gmap(w_f, b_f).double()
    assert w.shape == (64, 3 + Cg + 1, 3, 3)
    out = torch.nn.functional.conv2d(torch.cat([rgb, g, a], 1), w, c0.bias, padding=1)
    assert float((out - ref).abs().max()) < 1e-5 * float(ref.abs().max())          # (the composed weights are stored in fp32)
    assert fnet._conv0_on_gmap(w_f, b_f) is fnet._conv0_on_gmap(w_f, b_f)            # cached while nothing changes
    w_f.mul_(2.0)
    assert not torch.equal(fnet._conv0_on_gmap(w_f, b_f).double(), w)                # ... and recomposed when a weight is written
