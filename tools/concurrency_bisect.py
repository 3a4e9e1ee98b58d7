#!/usr/bin/env python3
"""Which backward kernel of the render is not bit-stable when two streams run it at once (tools/concurrency_check.py found the forward
stable and the backward not): composite_bwd -> field_bwd -> ray_grad_reduce -> raygen_bwd, each compared with its solo result."""
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


def once():
    o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
    raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
    raw.retain_grad()
    rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, flags, 0.03)
    loss = (rgb ** 2).sum() + (feat ** 2).sum()
    loss.backward()
    return dict(g_raw=raw.grad.clone(), g_o=o.grad.clone(), g_d=d.grad.clone(), g_v=v.grad.clone())


solo = once()
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
bad = {k: 0 for k in solo}
for rep in range(15):
    outs = []
    for s in streams:
        with torch.cuda.stream(s):
            outs.append(once())
    torch.cuda.synchronize()
    for o in outs:
        for k in solo:
            if not torch.equal(o[k], solo[k]):
                bad[k] += 1
                if bad[k] == 1:
                    dif = (o[k] - solo[k]).abs()
                    print(k, "first difference: max", float(dif.max()), "of", float(solo[k].abs().max()), "in", int((dif > 0).sum()), "of", dif.numel(), "entries;",
                          "nan" if torch.isnan(o[k]).any() else "")
                    idx = (dif > 0).nonzero()
                    if idx.shape[1] == 3:
                        rays = idx[:, 0].unique().tolist()
                        print("   rays", rays[:24], "rows", idx[:, 1].unique().tolist()[:24], "samples", idx[:, 2].unique().tolist()[:40])
                    else:
                        print("   rays", idx[:, 0].unique().tolist()[:24])
print("different from solo (of 30):", bad)
