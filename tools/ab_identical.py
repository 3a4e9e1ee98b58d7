"""Digest of the fp16 two-part field kernels' outputs on fixed inputs, for A/B builds that must not change a bit.
    NEFES_HIP_LIB=<lib> python tools/ab_identical.py [Wd C]
prints one sha256 per output of the forward kernels (sigma-only raw; full raw + ReLU masks).  Run it once per
library (tools/ab_h3.sh builds side libraries) and compare the lines: a re-scheduling of the same arithmetic gives the same digests."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF

dev = torch.device('cuda')
Wd, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 16)
N, S = 1000, 77                       # a ragged last tile, S not a multiple of anything
torch.manual_seed(0)
fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).to(dev)
with torch.no_grad():                 # spread the activations over a few octaves so that the exponent picks differ per sample
    for i, p in enumerate(fine.parameters()):
        p.mul_(1.0 + 0.5 * ((i * 7) % 5))
pk = fine.packed()
g = torch.Generator(device='cpu').manual_seed(1)
o = (torch.randn(N, 3, generator=g) * 0.3).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].to(dev)


def dig(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


print("lib", os.environ.get("NEFES_HIP_LIB") or "shipped")
raw, _ = ops.field_fwd_x6(pk, L.FIELD_SIGMA, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=False)
print("sigma raw  ", dig(raw))
for mode, name in ((L.FIELD_FULL, "full"),):
    raw, m = ops.field_fwd_x6(pk, mode, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=True)
    print(f"{name} raw   ", dig(raw), " masks", dig(m), " finite", bool(torch.isfinite(raw).all()))
