"""Diagnostic (NEFES_STAMP build of field_fwd.hip): share of a workgroup's cycles parked in the ring's counted wait and barrier."""
import sys, torch
sys.path.insert(0, '/root/repo')
from nefes_amd import lib as L, ops
from nefes_amd.field import NeRFH_NFF
import ctypes as C
dev = torch.device('cuda')
N, S = 76800, 64
net = NeRFH_NFF('coarse', W=256, f_dim=16).requires_grad_(False).to(dev)
pk = net.packed()
g = torch.Generator(device='cpu').manual_seed(0)
o = (torch.randn(N, 3, generator=g) * 0.2).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
z = torch.sort(torch.rand(N, S, generator=g) * 4, -1)[0].to(dev)
for mode, name in ((L.FIELD_SIGMA, 'sigma'), (L.FIELD_STATIC, 'static')):
    raw = torch.empty(N, pk.n_raw(mode), S, device=dev)
    stats = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
    L.check(L.load().nefes_field_fwd(pk.desc, pk.blob.data_ptr(), mode, N, S, o.data_ptr(), d.data_ptr(), z.data_ptr(), None, None,
                                     d.data_ptr(), raw.data_ptr(), stats.data_ptr(), ops._stream()), "fwd")
    torch.cuda.synchronize()
    st = stats.view(256, 4).double()
    tot = st[:, 2].sum()
    print(name, "wait %.2f%%  barrier %.2f%%  cycles/tile %.0f (memtime ticks)" % (100 * st[:, 0].sum() / tot, 100 * st[:, 1].sum() / tot, float(tot / st[:, 3].sum())))

# bf16x6 sigma kernel (NEFES_STAMP build of field_fwd_x6.hip)
raw = torch.empty(N, 1, S, device=dev)
stats = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
L.check(L.load().nefes_field_fwd_x6(pk.desc, pk.blob.data_ptr(), L.FIELD_SIGMA, N, S, o.data_ptr(), d.data_ptr(), z.data_ptr(), None,
                                    None, raw.data_ptr(), stats.data_ptr(), ops._stream()), "fwd_x6")
torch.cuda.synchronize()
st = stats.view(256, 4).double()
tot = st[:, 2].sum()
print("sigma x6", "wait %.2f%%  barrier %.2f%%  cycles/tile %.0f" % (100 * st[:, 0].sum() / tot, 100 * st[:, 1].sum() / tot, float(tot / st[:, 3].sum())))
