#!/usr/bin/env python3
"""Bit-stability under co-residency of the paths tools/concurrency_check.py does not cover: the headline network's chain (8 x 256, C = 16,
64 + 128 samples: field FULL -> compositing -> backward), the hash-grid instances, and a train-mode step (TRAIN instances: forward
saving activations, dX chain, dW kernels) -- each run on one stream while a second stream runs field kernels, compared with its solo result."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nefes_amd import ops, lib as L
from nefes_amd.field import NeRFH_NFF
from nefes_amd.render import render
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def rays(N, S):
    ro, rd = mk(N, 3) * 0.1, torch.nn.functional.normalize(mk(N, 3), dim=-1)
    z = (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(dev)
    return ro, rd, z


def chain_of(Wd, C, N, S, grid=None):
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, in_channels_xyz=32 if grid is not None else 63, encode_appearance=True,
                     encode_transient=True).requires_grad_(False).to(dev)
    pk = fine.packed()
    ro, rd, z = rays(N, S)

    def chain():
        o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
        if grid is not None:
            raw = ops.FieldFromRaysHashGrid.apply(o, d, v, z, pk, L.FIELD_FULL, grid)
        else:
            raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
        rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, L.COMP_TRANSIENT, 0.03)
        ((rgb ** 2).sum() + (feat ** 2).sum() + (disp ** 2).sum()).backward()
        return torch.cat([rgb.detach().reshape(-1), feat.detach().reshape(-1), o.grad.reshape(-1), d.grad.reshape(-1), v.grad.reshape(-1)])
    return chain


def train_step_of():
    H = W = 120
    focal, Wd, C = 525.505 * 200 / 480, 128, 128
    torch.manual_seed(0)
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).to(dev)
    prm = [p for n, p in coarse.named_parameters() if not n.startswith(("fusion_net", "exposure_embedding"))]
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1024)
    kw = dict(network_query_fn=None, perturb=0., N_importance=0, N_samples=64, network_fn=coarse, network_fine=None,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    ro, rd, _ = ops.raygen_fwd(H, W, focal, bench.bench_pose().to(dev))
    target = torch.rand(H * W, 3, generator=g).to(dev)

    def step():
        rgb, _, _, _ = render(H, W, focal, rays=(ro, rd), near=0., far=4., **kw)
        loss = ((rgb - target) ** 2).mean()
        grads = torch.autograd.grad(loss, prm)
        return torch.cat([rgb.detach().reshape(-1)] + [g_.reshape(-1) for g_ in grads])
    return step


def neighbour_of():
    fine = NeRFH_NFF('fine', W=128, f_dim=128, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
    pk = fine.packed()
    ro, rd, z = rays(4800, 128)

    def nb():
        with torch.no_grad():
            return ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)
    return nb


nb = neighbour_of()
grid = ops.HashGrid(25.0, device=dev)
grid.table.mul_(3e3)
cases = [("headline chain (8 x 256, C = 16, 192 samples)", chain_of(256, 16, 3200, 192)),
         ("width 256 with 128 feature channels", chain_of(256, 128, 2400, 128)),
         ("hash grid inside the field kernels (configs[3])", chain_of(256, 16, 3200, 192, grid)),
         ("train-mode step (forward saving activations, dX, dW)", train_step_of())]
for name, fn in cases:
    solo = fn().clone()
    torch.cuda.synchronize()
    for other_name, other in (("a field forward", nb), ("itself", fn)):
        bad = 0
        for rep in range(10):
            with torch.cuda.stream(streams[0]):
                a = fn()
            with torch.cuda.stream(streams[1]):
                b = other()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(a, solo) else 1
        print(f"{name}, next to {other_name}: different from solo {bad} of 10")
