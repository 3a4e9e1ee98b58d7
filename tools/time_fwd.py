"""Time the fused forward kernels alone (sigma-only and full) at the BASELINE shape on a slice of the frame."""
import sys, torch
sys.path.insert(0, '/root/repo')
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
dev = torch.device('cuda')
Wd, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 16)
N, S = 76800, 192
fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk = fine.packed()
g = torch.Generator(device='cpu').manual_seed(0)
o = (torch.randn(N, 3, generator=g) * 0.2).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].to(dev)
for mode, name in ((L.FIELD_SIGMA, 'sigma'), (L.FIELD_FULL, 'full')):
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw, m = ops.field_fwd(pk, mode, N, S, rays_o=o, rays_d=d, z=z, viewdirs=d, want_masks=(mode == L.FIELD_FULL))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    mac = {('sigma', 256): 491264, ('full', 256): 665088, ('sigma', 128): 130944, ('full', 128): 184064}[(name, Wd)]
    print(f"{name}: {ms:.2f} ms  {2 * mac * N * S / ms / 1e9:.1f} TFLOP/s  checksum {float(raw.double().sum()):.6f}")
# bf16 split-product instances (six products)
for mode, name in ((L.FIELD_SIGMA, 'sigma'), (L.FIELD_FULL, 'full')):
    if not ops.x6_supported(pk, mode):
        continue
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw, m = ops.field_fwd_x6(pk, mode, N, S, o, d, z, viewdirs=d, want_masks=(mode == L.FIELD_FULL))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    mac = {('sigma', 256): 491264, ('full', 256): 665088, ('sigma', 128): 130944, ('full', 128): 184064}[(name, Wd)]
    print(f"{name} bf16x6: {ms:.2f} ms  {2 * mac * N * S / ms / 1e9:.1f} TFLOP/s (algorithmic fp32 FLOPs)  "
          f"checksum {float(raw.double().sum()):.6f}")
