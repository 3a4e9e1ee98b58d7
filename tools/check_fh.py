"""Factored-head kernels against the plain ones, piece by piece (round 5; csrc/field_fwd_h3.hip FH): raw outputs, feat = W_f g + b_f, and the
ray gradients for upstream gradients on the colour channels only / the feature channels only / each of the six sigma + transient channels
(how the transient path was found broken when d loss / d g entered G2 as a C operand or a VALU update: profiles/r05/README.md).
    python tools/check_fh.py          (GPU box; NEFES_HIP_LIB=<side library> for A/B builds)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
dev="cuda"
torch.manual_seed(0)
fine = NeRFH_NFF('fine', W=128, f_dim=128, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk = fine.packed(); pk_fh, w_f, w_f_t, b_f = fine.packed_fh()
N,S=37,64
g=torch.Generator().manual_seed(1)
o=(torch.randn(N,3,generator=g)*0.3).to(dev); d=torch.nn.functional.normalize(torch.randn(N,3,generator=g),dim=-1).to(dev)
z=torch.sort(torch.rand(N,S,generator=g)*4,-1)[0].to(dev)
raw_p, m_p = ops.field_fwd_x6(pk, L.FIELD_FULL, N, S, o, d, z, viewdirs=d, want_masks=True)
oo,dd,vv=(t.clone().requires_grad_() for t in (o,d,d))
raw_f = ops.FieldFromRaysFH.apply(oo,dd,vv,z,pk_fh)
rel=lambda a,b: float((a-b).abs().max()/b.abs().max())
print("rgb", rel(raw_f[:,:3], raw_p[:,:3]), "sigma/transient", rel(raw_f[:,68:], raw_p[:,131:]))
gch = raw_f[:,3:67]          # [N,64,S]
feat_from_g = torch.einsum('cf,nfs->ncs', w_f, gch) + b_f[None,:,None]
print("feat via g", rel(feat_from_g, raw_p[:,3:131]), "ones", float(raw_f[:,67].min()), float(raw_f[:,67].max()), "g min", float(gch.min()))
G_rgb=torch.randn(N,3,S,generator=g).to(dev); G_feat=torch.randn(N,128,S,generator=g).to(dev); G_tail=torch.randn(N,6,S,generator=g).to(dev)
Gp=torch.cat([G_rgb,G_feat,G_tail],1).contiguous()
dg=torch.einsum('cf,ncs->nfs', w_f, G_feat)
Gf=torch.cat([G_rgb,dg,torch.zeros(N,1,S,device=dev),G_tail],1).contiguous()
gp_p, gv_p = ops.field_bwd(pk, N, S, raw_p, Gp, m_p, rays_o=o, rays_d=d, z=z, viewdirs=d)
raw_f.backward(Gf)
go_p, gd_p, gvv_p = ops.ray_grad_reduce(N,S,z,gp_p,gv_p)
print("d o", rel(oo.grad, go_p), "d d", rel(dd.grad, gd_p), "d v", rel(vv.grad, gvv_p))
# components: only rgb upstream / only feat upstream / only tail
for name,(a,b,c) in {"rgb":(1,0,0),"feat":(0,1,0),"tail":(0,0,1)}.items():
    Gp=torch.cat([G_rgb*a,G_feat*b,G_tail*c],1).contiguous()
    Gf=torch.cat([G_rgb*a,dg*b,torch.zeros(N,1,S,device=dev),G_tail*c],1).contiguous()
    gp_p, gv_p = ops.field_bwd(pk, N, S, raw_p, Gp, m_p, rays_o=o, rays_d=d, z=z, viewdirs=d)
    oo2,dd2,vv2=(t.clone().requires_grad_() for t in (o,d,d))
    rf = ops.FieldFromRaysFH.apply(oo2,dd2,vv2,z,pk_fh); rf.backward(Gf)
    go_p, gd_p, gvv_p = ops.ray_grad_reduce(N,S,z,gp_p,gv_p)
    print(name, "d o", rel(oo2.grad, go_p), "d d", rel(dd2.grad, gd_p), "d v", rel(vv2.grad, gvv_p))
print("---- per tail channel; and the plain class-0 kernels on the C=0 blob")
raw0, m0 = ops.field_fwd_x6(pk_fh, L.FIELD_FULL, N, S, o, d, z, viewdirs=d, want_masks=True)     # [N, 9, S]
print("C=0 plain fwd vs fh: rgb", rel(raw0[:,:3], raw_f[:,:3].detach()), "tail", rel(raw0[:,3:], raw_f[:,68:].detach()), "masks equal", bool(torch.equal(m0, m_p)))
for ch in range(6):
    Gt=torch.zeros(N,6,S,device=dev); Gt[:,ch]=G_tail[:,ch]
    Gp=torch.cat([G_rgb*0,G_feat*0,Gt],1).contiguous()
    Gf=torch.cat([G_rgb*0,dg*0,torch.zeros(N,1,S,device=dev),Gt],1).contiguous()
    G0=torch.cat([G_rgb*0,Gt],1).contiguous()
    gp_p, gv_p = ops.field_bwd(pk, N, S, raw_p, Gp, m_p, rays_o=o, rays_d=d, z=z, viewdirs=d)
    gp_0, gv_0 = ops.field_bwd(pk_fh, N, S, raw0, G0, m0, rays_o=o, rays_d=d, z=z, viewdirs=d)
    oo2,dd2,vv2=(t.clone().requires_grad_() for t in (o,d,d))
    rf = ops.FieldFromRaysFH.apply(oo2,dd2,vv2,z,pk_fh); rf.backward(Gf)
    go_p, gd_p, gvv_p = ops.ray_grad_reduce(N,S,z,gp_p,gv_p)
    go_0, gd_0, gvv_0 = ops.ray_grad_reduce(N,S,z,gp_0,gv_0)
    print("tail ch", ch, "fh vs plain: d o", rel(oo2.grad, go_p), "d v", rel(vv2.grad, gvv_p), "| C=0 plain vs plain: d o", rel(go_0, go_p), "d v", rel(gvv_0, gvv_p))
