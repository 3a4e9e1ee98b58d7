"""Embeddings with fewer octaves than the kernels compute (host logic, no GPU): the reference's --reduce_embedding 0 / 1 and smaller
--multires (script/models/nerfh_nff.py:303-354) are PREFIXES of the 63 / 27-feature embeddings, so such a network runs on the same
kernels with zero weight columns appended (nefes_amd/field.py NeRFH_NFF._kernel_params).  Checked here on the nn.Module API: the
padded network on the full embedding equals the network itself on its own embedding, and shrink_grads inverts the padding."""
import pytest
import torch

from nefes_amd.field import NeRFH_NFF, get_embedder


@pytest.mark.parametrize("multires,views,mode,want", [(10, 4, -1, (63, 27)), (10, 4, 0, (33, 15)), (10, 4, 1, (3, 3)), (6, 2, -1, (39, 15)),
                                                      (8, 3, 0, (27, 9))])
def test_embedder_sizes_follow_the_reference_rule(multires, views, mode, want):
    fx, nx, _ = get_embedder(multires, 0, mode)
    fd, nd, _ = get_embedder(views, 0, mode)
    assert (nx, nd) == want
    x = torch.randn(5, 3)
    full = get_embedder(10, 0, -1)[0](x)
    assert torch.equal(fx(x), full[:, :nx])              # a prefix of the paper-default embedding, bit for bit
    assert torch.equal(fd(x), get_embedder(4, 0, -1)[0](x)[:, :nd])


def test_mode_2_is_refused_with_the_reason():
    with pytest.raises(NotImplementedError, match="without the epoch argument"):
        get_embedder(10, 0, 2)


def test_non_octave_bands_are_refused():
    """multires = 9 in mode 0: four bands spread over 2^0 .. 2^4 (nerfh_nff.py:311-312 with linspace :253) are not octaves."""
    with pytest.raises(NotImplementedError, match="not the octaves"):
        get_embedder(9, 0, 0)


@pytest.mark.parametrize("in_xyz,in_dir", [(33, 15), (3, 3), (39, 27), (63, 9)])
def test_padded_network_equals_the_network(in_xyz, in_dir):
    net = NeRFH_NFF('fine', W=128, in_channels_xyz=in_xyz, in_channels_dir=in_dir, encode_appearance=True, encode_transient=True, f_dim=16)
    assert net._supported()
    sd = dict(net.named_parameters())
    ksd = net._kernel_params(sd)
    assert ksd["xyz_encoding_1.0.weight"].shape == (128, 63) and ksd["xyz_encoding_5.0.weight"].shape == (128, 63 + 128)
    assert ksd["dir_encoding.0.weight"].shape == (64, 128 + 27) and ksd["transient_encoding.0.weight"].shape == (64, 128 + 27)
    big = NeRFH_NFF('fine', W=128, encode_appearance=True, encode_transient=True, f_dim=16)
    big.load_state_dict({k: v for k, v in ksd.items()}, strict=False)
    x, d = torch.randn(7, 3), torch.nn.functional.normalize(torch.randn(7, 3), dim=-1)
    ex, ed = get_embedder(10, 0, -1)[0](x), get_embedder(4, 0, -1)[0](d)
    ref = net(torch.cat([ex[:, :in_xyz], ed[:, :in_dir]], 1))
    got = big(torch.cat([ex, ed], 1))
    assert torch.allclose(got, ref, rtol=0, atol=1e-6)
    # gradients of the padded shapes shrink back to the parameters' shapes, keeping the columns that exist
    g = {k: torch.arange(v.numel(), dtype=torch.float32).reshape(v.shape) for k, v in ksd.items()}
    sh = net.shrink_grads(dict(g))
    for k, v in sd.items():
        assert sh[k].shape == v.shape, k
    w5 = g["xyz_encoding_5.0.weight"]
    assert torch.equal(sh["xyz_encoding_5.0.weight"], torch.cat([w5[:, :in_xyz], w5[:, 63:]], 1))
    wd = g["dir_encoding.0.weight"]
    assert torch.equal(sh["dir_encoding.0.weight"], wd[:, :128 + in_dir])


def test_unsupported_input_widths_still_fail_loudly():
    for bad in (dict(in_channels_xyz=40), dict(in_channels_dir=20), dict(in_channels_xyz=69)):
        net = NeRFH_NFF('fine', W=128, encode_appearance=True, encode_transient=True, f_dim=16, **bad)
        assert not net._supported()
