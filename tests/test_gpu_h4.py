"""EXPERIMENT: the sigma-only forward on v_mfma_f32_16x16x32_f16 (csrc/field_fwd_h4.hip) against the
float64 oracle and against the production fp16 kernel, on the same rays; and its kernel time next to the production kernel's."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import parity_log as P

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rays(N, S, seed=0, scale=0.3, zmax=4.0):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(N, 3, generator=g) * scale
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * zmax, -1)[0]
    return o, d, z


@pytest.mark.parametrize("Wd,C,layout", [(256, 16, "b"), (256, 16, "a"), (128, 128, "b")])
@pytest.mark.parametrize("N,S", [(300, 64), (1, 1), (7, 19)])
def test_h4_sigma_matches_the_oracle(N, S, Wd, C, layout, monkeypatch):
    monkeypatch.setenv("NEFES_H4_LAYOUT", layout)
    from nefes_amd import lib as L, ops
    from nefes_amd.field import NeRFH_NFF
    net = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    h4 = ops.H4Sigma(net)
    o, d, z = rays(N, S)
    got = h4.forward(N, S, o.to(DEV), d.to(DEV), z.to(DEV))[:, 0].cpu().double()
    prod = ops.field_fwd_x6(net.packed(), L.FIELD_SIGMA, N, S, o.to(DEV), d.to(DEV), z.to(DEV))
    prod = (prod[0] if isinstance(prod, tuple) else prod)[:, 0].cpu().double()
    p64 = O.make_field_params("coarse", Wd, C, dtype=torch.float64)
    pts = o.double()[:, None] + d.double()[:, None] * z.double()[..., None]
    e = O.freq_encode(pts.reshape(-1, 3).float().double(), 10) if False else O.freq_encode((o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3).double(), 10)
    ref = O.field_forward(p64, e, sigma_only=True)[:, 0].reshape(N, S)
    p32 = O.make_field_params("coarse", Wd, C)
    ref32 = O.field_forward(p32, O.freq_encode((o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3), 10), sigma_only=True)[:, 0].reshape(N, S).double()
    scale = float(ref.abs().max())
    e_hip, e_ref = float((got - ref).abs().max()) / scale, float((ref32 - ref).abs().max()) / scale
    P.check(f"h4_sigma[{Wd},{layout},{N},{S}]", "sigma", e_hip, e_ref, float((got - prod).abs().max()) / scale, tol=3e-6, factor=3.0)


@pytest.mark.parametrize("Wd,C", [(256, 16), (128, 128)])
def test_h4_sigma_speed_against_production(Wd, C):
    """Not an assertion on speed: records both kernel times (76 800 rays x 64 samples) in the parity log for profiles/."""
    from nefes_amd import lib as L, ops
    from nefes_amd.field import NeRFH_NFF
    net = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    h4 = ops.H4Sigma(net)
    N, S = 76800, 64
    o, d, z = (t.to(DEV) for t in rays(N, S))
    pk = net.packed()

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5
    t_prod = timed(lambda: ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o, d, z))
    t_h4 = timed(lambda: h4.forward(N, S, o, d, z))
    P.record(f"h4_sigma_speed[{Wd}]", "ms per launch (76800 x 64)", e_hip=t_h4, e_ref=t_prod, direct=t_prod / t_h4, bound=None)
    print(f"Wd={Wd} sigma-only forward, 76 800 rays x 64 samples: production (32x32x16) {t_prod:.2f} ms, 16x16x32 experiment {t_h4:.2f} ms")
    assert t_h4 > 0
