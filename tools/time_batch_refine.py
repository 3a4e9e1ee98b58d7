import sys, time, types, numpy as np, torch
sys.path.insert(0, '.')
from nefes_amd.field import NeRFH_NFF
from nefes_amd.refine import PoseRefiner
dev = "cuda"
C = 128
coarse = NeRFH_NFF('coarse', W=128, f_dim=C).requires_grad_(False).to(dev)
fine = NeRFH_NFF('fine', W=128, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, encode_hist=True)
kw = dict(network_query_fn=None, perturb=0., N_importance=64, N_samples=64, network_fn=coarse, network_fine=fine, use_viewdirs=True,
          white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for B, graph in ((1, False), (3, False), (3, True), (8, True)):
    t0 = T()
    r = PoseRefiner(kw, args, (240, 320, 262.75), 0., 4., tinyscale=4, upsample=True, graph=graph, device=dev, images=B)
    init = torch.eye(4, device=dev)[None].repeat(B, 1, 1); hist = torch.full((B, 10), 10., device=dev)
    target = torch.randn(B, C, 220, 300, device=dev)
    if B == 1: init, target = init[0], target[0]
    t1 = T(); r.refine(init, target, hist, 2); t2 = T(); r.refine(init, target, hist, 50); t3 = T()
    print(f"B={B} graph={graph}: construct {t1-t0:.2f} s, first refine(2) {t2-t1:.2f} s, refine(50) {t3-t2:.3f} s = {(t3-t2)/50/B*1e3:.3f} ms per image-iteration", flush=True)
