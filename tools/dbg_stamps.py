"""Debug (GPU): phase shares of the fused forward kernel from a stamped build (NEFES_HIP_LIB=.../libnefes_hip_stamp.so)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from nefes_amd import ops, lib as L
from nefes_amd.field import NeRFH_NFF
lib = L.load()
m = NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).cuda()
pk = m.packed()
N, S = 40000, 192
g = torch.Generator().manual_seed(1)
o = (torch.rand(N, 3, generator=g) - .5).cuda(); d = torch.randn(N, 3, generator=g); v = (d / d.norm(dim=-1, keepdim=True)).cuda(); d = d.cuda()
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].cuda()
buf = (ctypes.c_ulonglong * 8)()
for mode, masks in ((2, True), (2, False), (0, False)):
    ops.field_fwd(pk, mode, N, S if mode else 64, rays_o=o, rays_d=d, z=z if mode else z[:, :64].contiguous(), viewdirs=v, want_masks=masks); torch.cuda.synchronize()
    lib.nefes_debug_read_stamps(buf, 1)
    ops.field_fwd(pk, mode, N, S if mode else 64, rays_o=o, rays_d=d, z=z if mode else z[:, :64].contiguous(), viewdirs=v, want_masks=masks); torch.cuda.synchronize()
    lib.nefes_debug_read_stamps(buf, 1)
    tot = sum(buf[:5])
    names = ["loads+drain", "embed(sincos)", "bias_init(+stores before it)", "mma (incl. next two)", "act_store", "  of mma: counted vmcnt wait", "  of mma: barrier", "-"]
    print(f"mode={mode} masks={masks}: total stamped cycles {tot:.3e}")
    for n, x in zip(names, buf):
        if x: print(f"   {n:32s} {100.0 * x / tot:6.2f} %")
