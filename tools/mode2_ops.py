#!/usr/bin/env python3
"""Where an iteration of mode 2 (train_on_batch with the stand-in regression network, eager) spends GPU and host time: torch.profiler
table over 10 iterations of bench.refinement_loop(mode="2")'s refiner."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nefes_amd import refine as R

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
keep = {}
orig = R.PoseRefiner.refine_apr


def spy(self, *a, **k):
    keep["ref"] = self
    return orig(self, *a, **k)


R.PoseRefiner.refine_apr = spy
bench.refinement_loop(dev, iters=3, graph=False, mode="2")
ref = keep["ref"]
n = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ref.apr_loss_and_grad(); ref.apr_opt.step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per iteration (eager, wall)")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(n):
        ref.apr_loss_and_grad(); ref.apr_opt.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=70))
