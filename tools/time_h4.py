"""Sigma-only forward on 76 800 rays x 64 samples: the production kernel (32x32x16) and the 16x16x32 experiment, five launches each."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
net = NeRFH_NFF('coarse', W=256, f_dim=16).requires_grad_(False).cuda()
h4 = ops.H4Sigma(net)
pk = net.packed()
N, S = 76800, 64
g = torch.Generator().manual_seed(0)
o = (torch.randn(N, 3, generator=g) * 0.3).cuda(); d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].cuda()
for name, fn in (("production", lambda: ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o, d, z)), ("h4 layout b", lambda: h4.forward(N, S, o, d, z)),
                 ("h4 layout a", lambda: h4.forward(N, S, o, d, z))):
    os.environ["NEFES_H4_LAYOUT"] = "a" if name.endswith(" a") else "b"
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 5:.3f} ms")
