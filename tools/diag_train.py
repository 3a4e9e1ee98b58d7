import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_train as T
from oracle import ref_cpu as O
from nefes_amd import lib as L, train as TR
Wd, C, typ = 256, 16, sys.argv[1] if len(sys.argv) > 1 else "coarse"
torch.manual_seed(11)
N, S = 37, 24
mode = L.FIELD_STATIC if typ == "coarse" else L.FIELD_FULL
net = T._net(typ, Wd, C)
g = torch.Generator().manual_seed(2)
rays_o = torch.randn(N, 3, generator=g) * 0.3
rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
z = torch.sort(torch.rand(N, S, generator=g) * 3.5 + 0.2, -1)[0]
raw_t = TR.field_train(net, mode, rays_o.cuda(), rays_d.cuda(), rays_d.cuda(), z.cuda())
G = torch.randn(N, raw_t.shape[1], S, generator=g)
(raw_t * G.cuda()).sum().backward()
names = TR.param_names(net, mode)
res = {}
for dt in (torch.float64, torch.float32):
    p = T._oracle_params(net, names, dt)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]).to(dt)
    raw = O.query_field(p, pts, rays_d.to(dt), typ, typ == "fine", False)
    (raw * G.permute(0, 2, 1).to(dt)).sum().backward()
    res[dt] = p
sd = dict(net.named_parameters())
for n in names:
    print(f"{n:32s} hip-vs-f64 {T._relerr(sd[n].grad, res[torch.float64][n].grad):.2e}   ref32-vs-f64 {T._relerr(res[torch.float32][n].grad, res[torch.float64][n].grad):.2e}")
# ---- intermediate check: G_4 from G_5 ----
import nefes_amd.train as TRm
TRm.DEBUG = {}
for pp in net.parameters(): pp.grad = None
raw_t = TR.field_train(net, mode, rays_o.cuda(), rays_d.cuda(), rays_d.cuda(), z.cuda())
(raw_t * G.cuda()).sum().backward()
D = TRm.DEBUG
acts, dacts, off = D["acts"], D["dacts"], D["off"]
W = Wd
for l in range(8, 1, -1):
    Gl = dacts[:, off[2 + l - 1]:off[2 + l - 1] + W, :]
    pre = acts[:, off[2 + l - 2]:off[2 + l - 2] + W, :]
    wl = sd[f"xyz_encoding_{l}.0.weight"].detach()
    wl = wl[:, 63:] if l == 5 else wl
    exp = torch.einsum("oi,tos->tis", wl, Gl) * (pre > 0)
    got = dacts[:, off[2 + l - 2]:off[2 + l - 2] + W, :]
    print("G", l - 1, "from G", l, float((exp - got).abs().max() / exp.abs().max()), "nonfinite", int((~torch.isfinite(got)).sum()))
# ---- saved pre-activations vs a float64 forward ----
p64 = res[torch.float64]
pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]).double().reshape(-1, 3)
e = O.freq_encode(pts, 10)
h = e
M = pts.shape[0]
for l in range(1, 9):
    if l == 5:
        h = torch.cat([e, h], 1)
    pre = torch.nn.functional.linear(h, p64[f"xyz_encoding_{l}.0.weight"].detach(), p64[f"xyz_encoding_{l}.0.bias"].detach())
    got = acts[:, off[2 + l - 1]:off[2 + l - 1] + W, :].permute(0, 2, 1).reshape(-1, W)[:M].cpu().double()
    flips = ((got > 0) != (pre > 0))
    print("pre", l, "maxabs diff", float((got - pre).abs().max()), "sign flips", int(flips.sum()), "|pre| at flips", pre[flips].abs().tolist()[:5])
    h = torch.relu(pre)
