"""Every GPU launch of ONE iteration of mode 2 (bench.refinement_loop(mode="2")'s refiner), in start order with durations and the gap
to the previous launch's end: first eager (operator names), then one replay of the captured graph (what the loop actually runs)."""
import os, sys
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


def show(fn, title):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    print(f"---- {title}: {len(ev)} launches")
    tot, last_end = 0.0, None
    for i, e in enumerate(ev):
        gap = 0.0 if last_end is None else e.time_range.start - last_end
        last_end = e.time_range.end
        tot += e.device_time
        print(f"{i:3d} {e.device_time:8.1f} us  gap {gap:7.1f}  {e.name[:100]}")
    print(f"kernel time {tot:.1f} us, first start to last end {ev[-1].time_range.end - ev[0].time_range.start:.1f} us")


for graph in (False, True):
    bench.refinement_loop(dev, iters=3, graph=graph, mode="2")
    ref = keep["ref"]
    if graph:
        show(lambda: ref.replay(apr=True), "one replay of the captured iteration")
    else:
        show(lambda: ref._apr_iteration(), "one eager iteration")
