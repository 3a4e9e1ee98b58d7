"""fp16 two-part split-product instances (csrc/field_h3.h, field_fwd_h3.hip, field_bwd_h3.hip; nefes_amd.ops.SPLIT = "h3", the
default) against the float64 oracle, the fp32-MFMA kernels and the bf16x6 kernels.

Every product runs on v_mfma_f32_32x32x16_f16 as hh + hl + lh of (hi, lo) fp16 pairs of power-of-two scaled operands; the bar
is the one the fp32 and bf16x6 kernels meet: no further from the float64 truth than the reference's own fp32 arithmetic.
The scale bookkeeping (per-sample exponents, per-matrix weight exponents, common exponents of products that share
accumulators) is exercised with tiny and large coordinates and with weights rescaled by 1e-3 .. 1e3 per layer."""
import pytest
import torch

from oracle import ref_cpu as O
from tests import parity_log as P

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _fp16_instances():
    from nefes_amd import ops
    old, ops.SPLIT = ops.SPLIT, "h3"
    old_use, ops.USE_X6 = ops.USE_X6, True
    yield
    ops.SPLIT, ops.USE_X6 = old, old_use


def _rays(N, S, seed, spread=0.3, zmax=3.8):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(N, 3, generator=g) * spread
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * zmax + 0.1, -1)[0]
    return o, d, z, g


def _net(typ, Wd=256, C=16, xyz=63):
    from nefes_amd.field import NeRFH_NFF
    if typ == "coarse":
        return NeRFH_NFF('coarse', W=Wd, f_dim=C, in_channels_xyz=xyz).requires_grad_(False).to(DEV)
    return NeRFH_NFF('fine', W=Wd, f_dim=C, in_channels_xyz=xyz, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)


def _per_channel_err(got_nrs, ref_nsr):
    sc = ref_nsr.abs().amax((0, 1)).clamp_min(1e-30)
    return float(((got_nrs.permute(0, 2, 1).cpu().double() - ref_nsr).abs().amax((0, 1)) / sc).max())


def _timer_keys(fn):
    from nefes_amd import ops
    ops.TIMERS = {}
    try:
        out = fn()
        return out, set(ops.TIMERS)
    finally:
        ops.TIMERS = None


@pytest.mark.parametrize("N,S", [(37, 64), (1, 5), (300, 64)])
def test_sigma_h3_vs_oracle(N, S):
    from nefes_amd import lib as L
    from nefes_amd import ops
    net = _net("coarse")
    with torch.no_grad():                                           # sigma spanning [0, 50]: the "surface" scene of SURVEY §8d
        net.static_sigma[0].weight.mul_(40.)
        net.static_sigma[0].bias.mul_(40.)
    pk = net.packed()
    o, d, z, _ = _rays(N, S, 7)
    (h3, _), keys = _timer_keys(lambda: ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o.to(DEV), d.to(DEV), z.to(DEV)))
    assert keys == {"field_fwd[sigma,h3]"}, keys                     # the fp16 instance really ran
    f32, _ = ops.field_fwd(pk, L.FIELD_SIGMA, N, S, rays_o=o.to(DEV), rays_d=d.to(DEV), z=z.to(DEV))
    p = {k: v.detach().cpu().double() for k, v in net.named_parameters() if not k.startswith(("fusion", "exposure"))}
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field(p, pts.double(), None, "coarse", False, True)[..., 0]
    ref32 = O.query_field({k: v.float() for k, v in p.items()}, pts, None, "coarse", False, True)[..., 0]
    sc = float(ref.abs().max())
    e_h3 = float((h3[:, 0].cpu().double() - ref).abs().max()) / sc
    e_f32 = float((f32[:, 0].cpu().double() - ref).abs().max()) / sc
    e_ref = float((ref32.double() - ref).abs().max()) / sc
    print(f"[h3] sigma vs float64: fp16x3 {e_h3:.2e}  fp32-MFMA {e_f32:.2e}  torch fp32 {e_ref:.2e}")
    P.record(f"h3_sigma[{N},{S}]", "sigma", e_hip=e_h3, e_ref=e_ref, direct=e_f32, bound=max(2e-6, 3 * e_ref))
    assert e_h3 <= max(2e-6, 3 * e_ref)


@pytest.mark.parametrize("Wd,C,N,S", [(256, 16, 41, 24), (256, 16, 300, 64), (128, 128, 61, 32)])
def test_full_h3_outputs_masks_and_backward(Wd, C, N, S):
    """FULL mode: all raw channels against the float64 oracle; ReLU-mask words against the fp32 kernel's (identical up to
    pre-activations within rounding of zero); backward-to-inputs against the fp32-MFMA backward on the same forward state."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    net = _net("fine", Wd, C)
    pk = net.packed()
    o, d, z, g = _rays(N, S, 9)
    od, dd, zd = o.to(DEV), d.to(DEV), z.to(DEV)
    (h3, m3), keys = _timer_keys(lambda: ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, od, dd, zd, viewdirs=dd, want_masks=True))
    assert keys == {"field_fwd[full,h3]"}, keys
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=od, rays_d=dd, z=zd, viewdirs=dd, want_masks=True)
    p = {k: v.detach().cpu().double() for k, v in net.named_parameters()}
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field(p, pts.double(), d.double(), "fine", True, True)
    ref32 = O.query_field({k: v.float() for k, v in p.items()}, pts, d, "fine", True, True)
    e_h3, e_f32, e_ref = _per_channel_err(h3, ref), _per_channel_err(f32, ref), _per_channel_err(ref32.permute(0, 2, 1), ref)
    print(f"[h3] Wd={Wd}: raw ({9 + C} ch) vs float64: fp16x3 {e_h3:.2e}  fp32-MFMA {e_f32:.2e}  torch fp32 {e_ref:.2e}")
    P.record(f"h3_full[{Wd},{C},{N},{S}]", "raw (worst channel)", e_hip=e_h3, e_ref=e_ref, direct=e_f32, bound=max(3e-6, 3 * e_ref))
    assert e_h3 <= max(3e-6, 3 * e_ref)
    words = 8 * (Wd // 64) + 4 * (Wd // 128)
    n32 = (N * S) // 32
    a, b = m3.view(-1, words, 64)[:n32], m32.view(-1, words, 64)[:n32]
    diff = int(sum(bin(int(v) & 0xffffffff).count("1") for v in (a ^ b).flatten().cpu().tolist() if v))
    print(f"[h3] ReLU-mask bits differing from the fp32 kernel: {diff} of {a.numel() * 32}")
    assert diff <= max(4, a.numel() * 32 // 100000)
    # backward on the SAME forward state (fp32 kernel's outputs and masks)
    G = torch.randn(N, 9 + C, S, generator=g).to(DEV)
    out = {}
    for split in ("f32", "h3"):
        ops.SPLIT = split
        (out[split]), keys = _timer_keys(lambda: ops.field_bwd(pk, N, S, f32, G, m32, rays_o=od, rays_d=dd, z=zd, viewdirs=dd))
        assert keys == {"field_bwd[h3]" if split == "h3" else "field_bwd"}, keys
    ops.SPLIT = "h3"
    for a_, b_, name in ((out["h3"][0], out["f32"][0], "g_pts"), (out["h3"][1], out["f32"][1], "g_viewdirs")):
        e = float((a_ - b_).abs().max() / b_.abs().max())
        print(f"[h3] backward {name}: fp16x3 vs fp32-MFMA {e:.2e}")
        P.record(f"h3_full[{Wd},{C},{N},{S}]", f"backward {name} vs fp32-MFMA kernel", e_hip=e, e_ref=None, bound=1e-5)
        assert e < 1e-5, name


def test_h3_hashgrid_encoding_instance():
    """The external-32-feature (hash grid) instances: forward against the fp32-MFMA kernel, backward to the encoding."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    N, S = 40, 48
    net = _net("fine", 256, 16, xyz=32)
    pk = net.packed()
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    enc = (torch.randn(N * S, 32, generator=g) * 0.5)
    enc[: S] *= 1e-4                                                # one ray with a tiny encoding (tcnn's 1e-4 init scale)
    enc = enc.to(DEV)
    (h3, m3), keys = _timer_keys(lambda: ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, xyz_enc=enc, viewdirs=d, want_masks=True))
    assert keys == {"field_fwd[full,h3]"}
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, xyz_enc=enc, viewdirs=d, want_masks=True)
    sc = f32.abs().amax((0, 2), keepdim=True).clamp_min(1e-30)
    e = float(((h3 - f32).abs() / sc).max())
    print(f"[h3] hash-grid instance: raw vs fp32-MFMA kernel {e:.2e}")
    assert e < 4e-6
    G = torch.randn(N, 25, S, generator=g).to(DEV)
    ge3, gv3 = ops.field_bwd(pk, N, S, f32, G, m32, viewdirs=d)
    ops.SPLIT = "f32"
    ge, gv = ops.field_bwd(pk, N, S, f32, G, m32, viewdirs=d)
    assert float((ge3 - ge).abs().max() / ge.abs().max()) < 1e-5 and float((gv3 - gv).abs().max() / gv.abs().max()) < 1e-5


@pytest.mark.parametrize("N,S,seed", [(1, 1, 0), (7, 33, 1), (129, 192, 2), (3, 256, 3), (1000, 5, 4)])
def test_h3_ragged_shapes_against_fp32_kernel(N, S, seed):
    from nefes_amd import lib as L
    from nefes_amd import ops
    net = _net("fine")
    pk = net.packed()
    g = torch.Generator().manual_seed(100 + seed)
    o = (torch.randn(N, 3, generator=g) * 0.5).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 6.0, -1)[0].to(DEV)
    G = torch.randn(N, 25, S, generator=g).to(DEV)
    h3, m3 = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, o, d, z, viewdirs=d, want_masks=True)
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=True)
    sc = f32.abs().amax((0, 2), keepdim=True).clamp_min(0.1 * float(f32.abs().max()))
    assert float(((h3 - f32).abs() / sc).max()) < 4e-6
    s3, _ = ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o, d, z)
    assert float((s3[:, 0] - f32[:, 19]).abs().max() / f32[:, 19].abs().max()) < 4e-6        # sigma-only instance, same net
    a = ops.field_bwd(pk, N, S, f32, G, m32, rays_o=o, rays_d=d, z=z, viewdirs=d)
    ops.SPLIT = "f32"
    b = ops.field_bwd(pk, N, S, f32, G, m32, rays_o=o, rays_d=d, z=z, viewdirs=d)
    for x, y in zip(a, b):
        assert float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)) < 1e-5


@pytest.mark.parametrize("case", ["tiny_and_far", "weights_1e3_1e-3", "dead_layer"])
def test_h3_scale_bookkeeping_under_stress(case):
    """fp16 has five exponent bits: the kernels rescale every operand per sample and product.  Coordinates from 1e-4 to 20,
    weights rescaled per layer by 1e3 / 1e-3 (activations spanning 1e-9 .. 1e9 through the trunk), a layer whose units are all
    dead: outputs stay within fp32-level error of the float64 oracle per channel and sample scale, no inf / nan."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    N, S = 48, 32
    net = _net("fine")
    o, d, z, g = _rays(N, S, 21)
    with torch.no_grad():
        if case == "tiny_and_far":
            o[:16] *= 1e-4
            z[:16] *= 1e-4
            o[16:32] = o[16:32] * 30 + 5.
        elif case == "weights_1e3_1e-3":
            for i, f in zip(range(1, 9), (1e3, 1e-3, 1e3, 1e3, 1e-3, 1e-3, 1e3, 1e-3)):
                getattr(net, f"xyz_encoding_{i}")[0].weight.mul_(f)
                getattr(net, f"xyz_encoding_{i}")[0].bias.mul_(f if i > 1 else 1.)
            net.transient_encoding[2].weight.mul_(1e3)
            net.dir_encoding[0].weight.mul_(1e-2)
        else:
            net.xyz_encoding_3[0].bias.fill_(-1e3)                  # every unit of layer 3 negative: relu output all zero
    net.invalidate_packed()
    pk = net.packed()
    od, dd, zd = o.to(DEV), d.to(DEV), z.to(DEV)
    h3, m3 = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, od, dd, zd, viewdirs=dd, want_masks=True)
    assert torch.isfinite(h3).all()
    p = {k: v.detach().cpu().double() for k, v in net.named_parameters()}
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field(p, pts.double(), d.double(), "fine", True, True)
    ref32 = O.query_field({k: v.float() for k, v in p.items()}, pts, d, "fine", True, True)
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=od, rays_d=dd, z=zd, viewdirs=dd, want_masks=True)
    e_h3, e_ref, e_f32 = _per_channel_err(h3, ref), _per_channel_err(ref32.permute(0, 2, 1), ref), _per_channel_err(f32, ref)
    print(f"[h3/{case}] raw vs float64 (worst channel, relative to that channel's maximum): fp16x3 {e_h3:.2e}  fp32-MFMA {e_f32:.2e}  "
          f"torch fp32 {e_ref:.2e}")
    # the rescaled network has a feature channel that nearly cancels (|max| 1e-3 against a bias of 6e-2): every kernel -- fp32
    # MFMA, bf16x6, fp16x3 -- is ~1e-5 of THAT channel's maximum away there; the fp16 kernel must not be worse than the fp32 one
    bound = max(3e-6, 3 * e_ref, 1.5 * e_f32)
    P.record(f"h3_stress[{case}]", "raw (worst channel)", e_hip=e_h3, e_ref=e_ref, direct=e_f32, bound=bound)
    assert e_h3 <= bound
    G = torch.randn(N, 25, S, generator=g).to(DEV)
    a = ops.field_bwd(pk, N, S, f32, G, m32, rays_o=od, rays_d=dd, z=zd, viewdirs=dd)
    ops.SPLIT = "f32"
    b = ops.field_bwd(pk, N, S, f32, G, m32, rays_o=od, rays_d=dd, z=zd, viewdirs=dd)
    for x, y, name in zip(a, b, ("g_pts", "g_viewdirs")):
        assert torch.isfinite(x).all()
        e = float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
        print(f"[h3/{case}] backward {name} vs fp32-MFMA {e:.2e}")
        assert e < 2e-5, name


def test_repacked_network_keeps_the_fp16_instances():
    """nefes_pack_device with the fp16 plan (the default) refreshes the fp16 streams: a re-packed network stays on the fp16
    instances and sees the new weights."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    net = _net("fine").requires_grad_(True)
    pk = net.packed()
    o, d, z, _ = _rays(9, 16, 5)
    args = (pk, L.FIELD_FULL, 9, 16, o.to(DEV), d.to(DEV), z.to(DEV))
    (a, _), keys = _timer_keys(lambda: ops.field_fwd_x6(*args, viewdirs=d.to(DEV)))
    assert keys == {"field_fwd[full,h3]"}
    with torch.no_grad():
        net.xyz_encoding_2[0].weight.mul_(1.5)
    assert net.packed() is pk and pk.h3_valid and pk.generation == 1
    (b, _), keys = _timer_keys(lambda: ops.field_fwd_x6(*args, viewdirs=d.to(DEV)))
    assert keys == {"field_fwd[full,h3]"}
    assert float((a - b).abs().max()) > 1e-4                        # the new weights are in use
    ops_split = ops.SPLIT
    try:
        ops.SPLIT = "x6"
        (c, _), keys = _timer_keys(lambda: ops.field_fwd_x6(*args, viewdirs=d.to(DEV)))
    finally:
        ops.SPLIT = ops_split
    assert keys == {"field_fwd[full,x6]"}
    assert float((c - b).abs().max() / c.abs().max()) < 1e-5        # ... and they are the same weights the bf16x6 stream holds


def test_repacked_network_falls_back_to_bf16x6(monkeypatch):
    """A re-pack WITHOUT the fp16 plan (ops.REPACK_H3 = False) does not write the fp16 streams: such a network must use the
    bf16x6 instances, never stale fp16 weights."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    monkeypatch.setattr(ops, "REPACK_H3", False)
    net = _net("fine").requires_grad_(True)                         # trainable: parameter updates re-pack on the device
    pk = net.packed()
    o, d, z, _ = _rays(9, 16, 5)
    args = (pk, L.FIELD_FULL, 9, 16, o.to(DEV), d.to(DEV), z.to(DEV))
    (a, _), keys = _timer_keys(lambda: ops.field_fwd_x6(*args, viewdirs=d.to(DEV)))
    assert keys == {"field_fwd[full,h3]"}
    with torch.no_grad():
        net.xyz_encoding_2[0].weight.mul_(1.5)                      # in place on the parameter: version bump -> device re-pack
    pk2 = net.packed()
    assert pk2 is pk and not pk.h3_valid
    (b, _), keys = _timer_keys(lambda: ops.field_fwd_x6(*args, viewdirs=d.to(DEV)))
    assert keys == {"field_fwd[full,x6]"}
    assert float((a - b).abs().max()) > 1e-4                        # the new weights are in use
    net.requires_grad_(False)                                       # frozen: the next value change goes through the host packer
    with torch.no_grad():
        net.xyz_encoding_2[0].weight.mul_(1.0000001)
    pk3 = net.packed()
    assert pk3 is not pk and pk3.h3_valid
    (c, _), keys = _timer_keys(lambda: ops.field_fwd_x6(pk3, *args[1:], viewdirs=d.to(DEV)))
    assert keys == {"field_fwd[full,h3]"}
    assert float((c - b).abs().max() / b.abs().max()) < 1e-5
