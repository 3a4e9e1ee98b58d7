import sys, torch
sys.path.insert(0, '.')
from oracle import ref_cpu as O
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
DEV="cuda"
N,S=48,32
net=NeRFH_NFF('fine', W=256, f_dim=16, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
g=torch.Generator().manual_seed(21)
o=torch.randn(N,3,generator=g)*0.3; d=torch.nn.functional.normalize(torch.randn(N,3,generator=g),dim=-1); z=torch.sort(torch.rand(N,S,generator=g)*3.8+0.1,-1)[0]
with torch.no_grad():
    for i,f in zip(range(1,9),(1e3,1e-3,1e3,1e3,1e-3,1e-3,1e3,1e-3)):
        getattr(net,f"xyz_encoding_{i}")[0].weight.mul_(f); getattr(net,f"xyz_encoding_{i}")[0].bias.mul_(f if i>1 else 1.)
    net.transient_encoding[2].weight.mul_(1e3); net.dir_encoding[0].weight.mul_(1e-2)
net.invalidate_packed(); pk=net.packed()
od,dd,zd=o.to(DEV),d.to(DEV),z.to(DEV)
p={k:v.detach().cpu().double() for k,v in net.named_parameters()}
pts=o[:,None,:]+d[:,None,:]*z[...,None]
ref=O.query_field(p,pts.double(),d.double(),"fine",True,True)
ref32=O.query_field({k:v.float() for k,v in p.items()},pts,d,"fine",True,True)
sc=ref.abs().amax((0,1)).clamp_min(1e-30)
def err(t): return ((t.permute(0,2,1).cpu().double()-ref).abs().amax((0,1))/sc)
out={}
for split in ("h3","x6","f32"):
    ops.SPLIT=split
    if split=="f32": r,_=ops.field_fwd(pk,L.FIELD_FULL,N,S,rays_o=od,rays_d=dd,z=zd,viewdirs=dd)
    else: r,_=ops.field_fwd_x6(pk,L.FIELD_FULL,N,S,od,dd,zd,viewdirs=dd)
    out[split]=err(r)
e32=((ref32.double()-ref).abs().amax((0,1))/sc)
torch.set_printoptions(precision=2, sci_mode=True, linewidth=250)
print("chan max", sc)
for k,v in out.items(): print(k, v)
print("t32", e32)
