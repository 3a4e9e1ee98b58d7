"""bf16x6 split-product instances of the fused forward (field_fwd_x6.hip) against the fp32-MFMA kernels and the float64 oracle.

The hidden 256x256 products run on v_mfma_f32_32x32x16_bf16 with exact hi/mid/lo bf16 triples and six cross terms; the
bar is the same as for the fp32 kernels: no further from the float64 truth than the reference's own fp32 arithmetic.
"""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


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
    x6 = ops.field_fwd_sigma_x6(pk, N, S, o.to(DEV), d.to(DEV), z.to(DEV))
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
