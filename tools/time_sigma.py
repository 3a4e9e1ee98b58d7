"""Kernel time of the sigma-only forward on the stamp tool's inputs (random rays) at both widths, and on a smooth ray bundle."""
import sys, torch
sys.path.insert(0, '.')
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
L.load()
N, S = 76800, 64
g = torch.Generator().manual_seed(0)
for WD, CF in ((256, 16), (128, 128)):
    net = NeRFH_NFF('coarse', W=WD, f_dim=CF).requires_grad_(False).cuda()
    pk = net.packed()
    for name, scale, zmax in (("random rays, |o|~0.3, z<4", 0.3, 4.0), ("tiny scene, z<0.5", 0.03, 0.5)):
        o = (torch.randn(N, 3, generator=g) * scale).cuda(); d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
        z = torch.sort(torch.rand(N, S, generator=g) * zmax, -1)[0].cuda()
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, o, d, z); e1.record(); torch.cuda.synchronize()
        print(WD, name, f"{e0.elapsed_time(e1):.2f} ms")
