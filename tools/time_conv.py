"""csrc/conv.hip at FusionNet's layer shapes (60x80 image): microseconds per launch, back to back on one stream (steady state), and
torch's conv2d (MIOpen) beside it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nefes_amd import ops  # noqa: E402

dev = "cuda"
for name, cin, cout, k, mask in (("L1 fwd", 131, 64, 3, False), ("L2 fwd", 64, 64, 3, False), ("L4 fwd", 64, 128, 5, False),
                                 ("L4 dgrad", 128, 64, 5, False), ("L2 dgrad", 64, 64, 3, True), ("L1 dgrad", 64, 131, 3, True)):
    x = torch.randn(1, cin, 60, 80, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    m = torch.randn(1, cin, 60, 80, device=dev) if mask else None
    wp = ops._pack_conv(w, False)
    for fn, tag in ((lambda: ops._conv2d_same(x, wp, cout, k, b, True, mask=m), "hip"),
                    (lambda: torch.relu(torch.nn.functional.conv2d(x, w, b, padding=k // 2)), "torch")):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:9s} {cin:3d}->{cout:3d} {k}x{k} {tag:5s} {e0.elapsed_time(e1) / 300 * 1e3:7.1f} us per call")
