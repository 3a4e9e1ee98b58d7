import sys, torch
sys.path.insert(0, '/root/repo')
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
from oracle import ref_cpu as O
dev='cuda'
torch.manual_seed(0)
net = NeRFH_NFF('fine', W=256, f_dim=16, in_channels_xyz=32, encode_appearance=True, encode_transient=True).requires_grad_(False)
N, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 20)
g = torch.Generator().manual_seed(1)
enc = torch.randn(N, S, 32, generator=g) * 0.5
vd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
vd64 = vd.double().requires_grad_()
G = torch.randn(N, 25, S, generator=g)
# reference (fp64 torch module forward on pre-embedded inputs)
net64 = NeRFH_NFF('fine', W=256, f_dim=16, in_channels_xyz=32, encode_appearance=True, encode_transient=True).double()
e64 = enc.double().reshape(-1, 32).requires_grad_()
de = O.freq_encode(vd64, 4)[:, None, :].expand(N, S, 27).reshape(-1, 27)
out = net64(torch.cat([e64, de], 1))           # [M, 25]
(out.reshape(N, S, 25) * G.permute(0, 2, 1).double()).sum().backward()
ref = e64.grad.reshape(N, S, 32)
netg = net.to(dev)
for trial in range(3):
    e = enc.to(dev).requires_grad_()
    vg = vd.to(dev).requires_grad_()
    raw_t = ops.FieldFromEncoding.apply(e, vg, netg.packed(), L.FIELD_FULL)
    (raw_t * G.to(dev)).sum().backward()
    err = (e.grad.cpu().double() - ref).abs().max() / ref.abs().max()
    fe = (raw_t.permute(0, 2, 1).reshape(-1, 25).cpu().double() - out.detach()).abs().max()
    bad = ((e.grad.cpu().double() - ref).abs() > 1e-3 * ref.abs().max())
    print('g_view rel err', float((vg.grad.cpu().double() - vd64.grad).abs().max() / vd64.grad.abs().max()))
    print('trial', trial, 'fwd err', float(fe), 'g_enc rel err', float(err), 'bad elements', int(bad.sum()), 'of', bad.numel(), 'bad feature idx', sorted(set(bad.nonzero()[:, 2].tolist()))[:40])
