import sys, torch
sys.path.insert(0, '.')
from nefes_amd import ops, lib as L
from nefes_amd.field import NeRFH_NFF
Wd, C, N, S, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).cuda()
pk = m.packed()
g = torch.Generator().manual_seed(1)
o = (torch.rand(N, 3, generator=g) - .5).cuda(); d = torch.randn(N, 3, generator=g); v = (d / d.norm(dim=-1, keepdim=True)).cuda(); d = d.cuda()
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].cuda()
for r in range(reps):
    raw_t, masks = ops.field_fwd(pk, 2, N, S, rays_o=o, rays_d=d, z=z, viewdirs=v, want_masks=True)
    torch.cuda.synchronize(); print("fwd+masks ok", r, flush=True)
ref = raw_t.clone()
g_raw = torch.randn(N, raw_t.shape[1], S, generator=g).cuda()
for r in range(reps):
    gp, gv = ops.field_bwd(pk, N, S, raw_t, g_raw, masks, rays_o=o, rays_d=d, z=z, viewdirs=v)
    torch.cuda.synchronize(); print("bwd ok", r, float(gp.abs().max()), flush=True)
