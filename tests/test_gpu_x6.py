"""bf16x6 split-product instances of the fused forward (field_fwd_x6.hip) against the fp32-MFMA kernels and the float64 oracle.

The hidden 256x256 products run on v_mfma_f32_32x32x16_bf16 with exact hi/mid/lo bf16 triples and six cross terms; the
bar is the same as for the fp32 kernels: no further from the float64 truth than the reference's own fp32 arithmetic.
"""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _bf16x6_instances():
    """This module pins the bf16 split-product instances (nefes_amd.ops.SPLIT = "x6"); the default fp16 two-part instances
    have their own module, tests/test_gpu_h3.py."""
    from nefes_amd import ops
    old, ops.SPLIT = ops.SPLIT, "x6"
    yield
    ops.SPLIT = old


@pytest.mark.parametrize("N,S", [(37, 64), (1, 5), (300, 64)])
def test_sigma_x6_matches_fp32_kernel_and_oracle(N, S):
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    net = NeRFH_NFF('coarse', W=256, f_dim=16).requires_grad_(False).to(DEV)
    with torch.no_grad():                                           # sigma spanning [0, 50]: the "surface" scene of SURVEY §8d
        net.static_sigma[0].weight.mul_(40.)
        net.static_sigma[0].bias.mul_(40.)
    pk = net.packed()
    g = torch.Generator().manual_seed(7)
    o = torch.randn(N, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * 3.8 + 0.1, -1)[0]
    x6, _ = ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o.to(DEV), d.to(DEV), z.to(DEV))
    f32, _ = ops.field_fwd(pk, L.FIELD_SIGMA, N, S, rays_o=o.to(DEV), rays_d=d.to(DEV), z=z.to(DEV))
    p = {k: v.detach().cpu().double() for k, v in net.named_parameters() if not k.startswith(("fusion", "exposure"))}
    pts = (o[:, None, :] + d[:, None, :] * z[..., None])            # fp32 positions, as both kernels compute them
    ref = O.query_field(p, pts.double(), None, "coarse", False, True)[..., 0]          # [N,S] float64
    p32 = {k: v.float() for k, v in p.items()}
    ref32 = O.query_field(p32, pts, None, "coarse", False, True)[..., 0]
    sc = float(ref.abs().max())
    e_x6 = float((x6[:, 0].cpu().double() - ref).abs().max()) / sc
    e_f32 = float((f32[:, 0].cpu().double() - ref).abs().max()) / sc
    e_ref = float((ref32.double() - ref).abs().max()) / sc
    print(f"[x6] sigma vs float64: bf16x6 {e_x6:.2e}  fp32-MFMA {e_f32:.2e}  torch fp32 {e_ref:.2e}")
    assert e_x6 <= max(2e-6, 3 * e_ref)


@pytest.mark.parametrize("N,S", [(41, 24), (300, 64)])
def test_full_x6_outputs_and_masks(N, S):
    """FULL mode: all 25 raw channels against the float64 oracle, and the ReLU-mask words against the fp32 kernel's
    (identical up to pre-activations within rounding of zero), so the unchanged backward kernel follows either forward."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    net = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    pk = net.packed()
    g = torch.Generator().manual_seed(9)
    o = torch.randn(N, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * 3.8 + 0.1, -1)[0]
    od, dd, zd = o.to(DEV), d.to(DEV), z.to(DEV)
    x6, m6 = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, od, dd, zd, viewdirs=dd, want_masks=True)
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=od, rays_d=dd, z=zd, viewdirs=dd, want_masks=True)
    p = {k: v.detach().cpu().double() for k, v in net.named_parameters()}
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field(p, pts.double(), d.double(), "fine", True, True)               # [N,S,25] float64
    ref32 = O.query_field({k: v.float() for k, v in p.items()}, pts, d, "fine", True, True)
    sc = ref.abs().amax((0, 1)).clamp_min(1e-30)                                         # per channel
    e_x6 = float(((x6.permute(0, 2, 1).cpu().double() - ref).abs().amax((0, 1)) / sc).max())
    e_f32 = float(((f32.permute(0, 2, 1).cpu().double() - ref).abs().amax((0, 1)) / sc).max())
    e_ref = float(((ref32.double() - ref).abs().amax((0, 1)) / sc).max())
    print(f"[x6] raw (25 ch) vs float64: bf16x6 {e_x6:.2e}  fp32-MFMA {e_f32:.2e}  torch fp32 {e_ref:.2e}")
    assert e_x6 <= max(3e-6, 3 * e_ref)
    n_tiles32 = ((N * S + 127) // 128) * 4
    a = m6.view(n_tiles32, -1, 64)[: (N * S) // 32]                                     # whole 32-sample tiles only
    b = m32.view(n_tiles32, -1, 64)[: (N * S) // 32]
    diff_bits = int(sum(bin(int(v) & 0xffffffff).count("1") for v in (a ^ b).flatten().cpu().tolist() if v))
    total_bits = a.numel() * 32
    print(f"[x6] ReLU-mask bits differing from the fp32 kernel: {diff_bits} of {total_bits}")
    assert diff_bits <= max(4, total_bits // 100000)


def test_render_with_x6_forward_matches_oracle_gradient():
    """End to end at the headline shape per ray: render() with the bf16x6 forward kernels (ops.USE_X6) and the unchanged
    backward against the float64 oracle's pose gradient -- same bar as tests/test_gpu_parity.py."""
    import types
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render
    coarse = NeRFH_NFF('coarse', W=256, f_dim=16).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True)
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, N_samples=64, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
    H, W, f = 6, 8, 525.505 * 8 / 640.
    res = {}
    for use in (False, True):
        old, ops.USE_X6 = ops.USE_X6, use
        try:
            c2w = O.bench_pose().to(DEV).requires_grad_()
            rgb, disp, acc, ex = render(H, W, f, c2w=c2w, near=0., far=4., **kw)
            O.bench_loss(rgb, ex["feat_map"]).backward()
            res[use] = (rgb.detach().cpu(), ex["feat_map"].detach().cpu(), c2w.grad.cpu())
        finally:
            ops.USE_X6 = old

    def oracle(dt):
        pc = {k: v.detach().cpu().to(dt) for k, v in coarse.named_parameters() if not k.startswith(("fusion", "exposure"))}
        pf = {k: v.detach().cpu().to(dt) for k, v in fine.named_parameters()}
        c = O.bench_pose(dt).requires_grad_()
        rgb, _, _, ex = O.render(H, W, f, pc, pf, O.RenderCfg(N_samples=64, N_importance=128), c2w=c, near=0., far=4.)
        O.bench_loss(rgb, ex["feat_map"]).backward()
        return rgb.detach(), ex["feat_map"].detach(), c.grad

    r64, r32 = oracle(torch.float64), oracle(torch.float32)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(res[True][0], r64[0]) < 1e-4 and rel(res[True][1], r64[1]) < 1e-4
    e_x6, e_f32, e_ref = rel(res[True][2], r64[2]), rel(res[False][2], r64[2]), rel(r32[2], r64[2])
    print(f"[x6] d c2w vs float64: bf16x6 forward {e_x6:.2e}  fp32-MFMA forward {e_f32:.2e}  torch fp32 {e_ref:.2e}")
    assert e_x6 <= max(1e-4, 3 * e_ref)


def test_backward_x6_matches_fp32_backward():
    """nefes_field_bwd_x6 against nefes_field_bwd on identical forward outputs, masks and upstream gradients."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    N, S = 53, 40
    net = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    pk = net.packed()
    g = torch.Generator().manual_seed(13)
    o = (torch.randn(N, 3, generator=g) * 0.3).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 3.8 + 0.1, -1)[0].to(DEV)
    G = torch.randn(N, 25, S, generator=g).to(DEV)
    raw_t, masks = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=True)
    out = {}
    for use in (False, True):
        old, ops.USE_X6 = ops.USE_X6, use
        try:
            out[use] = ops.field_bwd(pk, N, S, raw_t, G, masks, rays_o=o, rays_d=d, z=z, viewdirs=d)
        finally:
            ops.USE_X6 = old
    for a, b, name in ((out[True][0], out[False][0], "g_pts"), (out[True][1], out[False][1], "g_viewdirs")):
        e = float((a - b).abs().max() / b.abs().max())
        print(f"[x6] backward {name}: bf16x6 vs fp32-MFMA {e:.2e}")
        assert e < 2e-5, name


def test_reference_default_shape_x6_forward():
    """Wd = 128, C = 128 (the reference's default shape): bf16x6 forward instances against the float64 oracle and the fp32
    kernel's mask words."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    N, S = 61, 32
    fine = NeRFH_NFF('fine', W=128, f_dim=128, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    pk = fine.packed()
    g = torch.Generator().manual_seed(17)
    o = torch.randn(N, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * 3.8 + 0.1, -1)[0]
    od, dd, zd = o.to(DEV), d.to(DEV), z.to(DEV)
    x6, m6 = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, od, dd, zd, viewdirs=dd, want_masks=True)
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=od, rays_d=dd, z=zd, viewdirs=dd, want_masks=True)
    s6, _ = ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, od, dd, zd)
    p = {k: v.detach().cpu().double() for k, v in fine.named_parameters()}
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field(p, pts.double(), d.double(), "fine", True, True)
    ref32 = O.query_field({k: v.float() for k, v in p.items()}, pts, d, "fine", True, True)
    sc = ref.abs().amax((0, 1)).clamp_min(1e-30)
    e_x6 = float(((x6.permute(0, 2, 1).cpu().double() - ref).abs().amax((0, 1)) / sc).max())
    e_ref = float(((ref32.double() - ref).abs().amax((0, 1)) / sc).max())
    e_sig = float((s6[:, 0].cpu().double() - ref[..., 3 + 128]).abs().max() / ref[..., 3 + 128].abs().max())
    print(f"[x6] Wd=128: raw (137 ch) vs float64: bf16x6 {e_x6:.2e}  torch fp32 {e_ref:.2e};  sigma-only {e_sig:.2e}")
    assert e_x6 <= max(3e-6, 3 * e_ref) and e_sig <= max(3e-6, 3 * e_ref)
    n32, words = (N * S) // 32, 8 * (128 // 64) + 4 * (128 // 128)                      # whole 32-sample tiles, mask words
    a, b = m6.view(-1, words, 64)[:n32], m32.view(-1, words, 64)[:n32]
    diff = int(sum(bin(int(v) & 0xffffffff).count("1") for v in (a ^ b).flatten().cpu().tolist() if v))
    assert diff <= 4, diff


@pytest.mark.parametrize("N,S,seed", [(1, 1, 0), (7, 33, 1), (129, 192, 2), (3, 256, 3), (1000, 5, 4)])
def test_x6_ragged_shapes_against_fp32_kernel(N, S, seed):
    """Single sample, ragged last tile, the maximum sample count per ray: bf16x6 outputs equal the fp32 kernel's to fp32
    rounding (both are within a few 1e-7 of float64, see above); forward FULL and backward."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    net = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    pk = net.packed()
    g = torch.Generator().manual_seed(100 + seed)
    o = (torch.randn(N, 3, generator=g) * 0.5).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 6.0, -1)[0].to(DEV)
    G = torch.randn(N, 25, S, generator=g).to(DEV)
    x6, m6 = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, o, d, z, viewdirs=d, want_masks=True)
    f32, m32 = ops.field_fwd(pk, L.FIELD_FULL, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=True)
    sc = f32.abs().amax((0, 2), keepdim=True).clamp_min(0.1 * float(f32.abs().max()))   # per channel, floored for tiny batches
    assert float(((x6 - f32).abs() / sc).max()) < 4e-6
    outs = {}
    for use in (False, True):
        old, ops.USE_X6 = ops.USE_X6, use
        try:
            outs[use] = ops.field_bwd(pk, N, S, f32, G, m32, rays_o=o, rays_d=d, z=z, viewdirs=d)
        finally:
            ops.USE_X6 = old
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < 2e-5
