#!/usr/bin/env python3
"""composite_bwd alone and field_bwd alone under two concurrent streams (see tools/concurrency_bisect.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd import ops, lib as L
from nefes_amd.field import NeRFH_NFF
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
Wd, C, N, S = 128, 128, 4800, 128
torch.manual_seed(0)
fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
pk = fine.packed()
g = torch.Generator().manual_seed(1)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
ro, rd = mk(N, 3) * 0.1, torch.nn.functional.normalize(mk(N, 3), dim=-1)
z = (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(dev)
flags = L.COMP_TRANSIENT
with torch.no_grad():
    raw0 = ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL).clone()
g_raw0 = torch.randn(raw0.shape, generator=g).to(dev) * 1e-2
torch.cuda.synchronize()


def comp_only():
    raw = raw0.clone().requires_grad_()
    rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, flags, 0.03)
    ((rgb ** 2).sum() + (feat ** 2).sum()).backward()
    return raw.grad.clone()


def field_only():
    o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
    raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
    gin = g_raw0.clone()
    raw.backward(gin)
    return torch.cat([o.grad, d.grad, v.grad], 1), gin


def field_fwd_only():
    with torch.no_grad():
        return ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL).clone()


streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for name, fa, fb in (("composite alone on both streams", comp_only, comp_only), ("field fwd+bwd alone on both streams", field_only, field_only),
                     ("composite on one stream, field forward on the other", comp_only, field_fwd_only),
                     ("composite on one stream, field fwd+bwd on the other", comp_only, field_only)):
    sa, sb = fa(), fb()
    torch.cuda.synchronize()
    first = lambda t: t[0] if isinstance(t, tuple) else t
    bad = [0, 0]
    inplace = 0
    for rep in range(15):
        with torch.cuda.stream(streams[0]):
            a = fa()
        with torch.cuda.stream(streams[1]):
            b = fb()
        torch.cuda.synchronize()
        bad[0] += 0 if torch.equal(first(a), first(sa)) else 1
        bad[1] += 0 if torch.equal(first(b), first(sb)) else 1
        if isinstance(b, tuple):
            inplace += 0 if torch.equal(b[1], g_raw0) else 1
    print(f"{name}: different from solo {bad} of 15 each; field backward changed its incoming gradient in place: {inplace}")


def chain():
    o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
    raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
    rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, flags, 0.03)
    ((rgb ** 2).sum() + (feat ** 2).sum()).backward()
    return torch.cat([o.grad, d.grad, v.grad], 1)


def chain_fwd_only():
    with torch.no_grad():
        raw = ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)
        rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, flags, 0.03)
        return torch.cat([rgb, feat], 1)


sa = chain()
torch.cuda.synchronize()
for name, fb in (("nothing", None), ("field forward", field_fwd_only), ("composite fwd+bwd", comp_only), ("field fwd+bwd", field_only),
                 ("chain forward only", chain_fwd_only), ("the chain", chain)):
    bad = 0
    for rep in range(15):
        with torch.cuda.stream(streams[0]):
            a = chain()
        if fb is not None:
            with torch.cuda.stream(streams[1]):
                b = fb()
        torch.cuda.synchronize()
        bad += 0 if torch.equal(a, sa) else 1
    print(f"chain on stream 0 with [{name}] on stream 1: different from solo {bad} of 15")


def chain_tapped():
    o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
    raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
    early = []
    raw.register_hook(lambda g_: early.append((g_.clone(), g_)))
    rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, flags, 0.03)
    g_rgb_in = []
    rgb.register_hook(lambda g_: g_rgb_in.append(g_.clone()))
    feat.register_hook(lambda g_: g_rgb_in.append(g_.clone()))
    ((rgb ** 2).sum() + (feat ** 2).sum()).backward()
    return dict(raw=raw.detach().clone(), rgb=rgb.detach().clone(), feat=feat.detach().clone(), g_maps=torch.cat([t.reshape(-1) for t in g_rgb_in]),
                g_raw_early=early[0][0], g_raw_late=early[0][1].clone(), g_rays=torch.cat([o.grad, d.grad, v.grad], 1))


sa = chain_tapped()
torch.cuda.synchronize()
bad = {k: 0 for k in sa}
for rep in range(20):
    with torch.cuda.stream(streams[0]):
        a = chain_tapped()
    with torch.cuda.stream(streams[1]):
        b = field_fwd_only()
    torch.cuda.synchronize()
    for k in sa:
        bad[k] += 0 if torch.equal(a[k], sa[k]) else 1
print("chain (tapped) with a field forward on the other stream, different from solo of 20:", bad)

found = 0
for rep in range(1500):
    with torch.cuda.stream(streams[0]):
        a = chain_tapped()
    with torch.cuda.stream(streams[1]):
        b = field_fwd_only()
    torch.cuda.synchronize()
    if not torch.equal(a["g_raw_early"], sa["g_raw_early"]):
        dif = (a["g_raw_early"] - sa["g_raw_early"]).abs()
        idx = (dif > 0).nonzero()
        r, row = int(idx[0, 0]), int(idx[0, 1])
        torch.set_printoptions(precision=5, linewidth=220)
        g3 = sa["g_maps"][3 * r:3 * r + 3].cpu()
        print(f"rep {rep}: {idx.shape[0]} entries, rows {idx[:, 1].unique().tolist()}, rays {idx[:, 0].unique().tolist()[:6]}; ray {r}: g_rgb {g3.tolist()}")
        so, co = sa["g_raw_early"][r, row, 64:128].cpu().reshape(16, 4), a["g_raw_early"][r, row, 64:128].cpu().reshape(16, 4)
        print("  concurrent / solo per lane (rows: lanes 48..63, columns k = 0..3):")
        print((co / so).t())
        print("  concurrent values k=0:", co[:, 0], " k=2:", co[:, 2])
        found += 1
        if found >= 4:
            break
