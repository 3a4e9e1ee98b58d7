"""FusionNet (4 convolutions through MIOpen) forward + backward to its inputs at the refinement shape, with MIOpen's default
(immediate-mode heuristics) and with torch.backends.cudnn.benchmark = True (MIOpen find mode: measured per shape at first call)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd.field import NeRFH_NFF  # noqa: E402

net = NeRFH_NFF('coarse', W=128, f_dim=128).requires_grad_(False).cuda()
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for B in (1, 8):
        rgb = torch.rand(B * 4800, 3, device='cuda', requires_grad=True)
        feat = torch.randn(B * 4800, 128, device='cuda', requires_grad=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, _, f = net.run_fusion_net(rgb, feat, 60, 80, B, per_image_norm=B > 1)
        f.sum().backward()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20):
            _, _, f = net.run_fusion_net(rgb, feat, 60, 80, B, per_image_norm=B > 1); f.sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"cudnn.benchmark={bench} B={B}: first call {t1 - t0:.2f} s, then {(t2 - t1) / 20 * 1e3:.3f} ms per fwd+bwd")
