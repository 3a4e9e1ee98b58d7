#!/usr/bin/env python3
"""Which Python lines of one EAGER refinement iteration launch torch's own small kernels (fills, copies, element-wise glue): the
iteration is a replayed graph in production, so each of them costs its ~4 us of execution, not a launch -- this lists them with the
source line that asked for them.  python tools/loop_ops.py [mode]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nefes_amd import refine as R

mode = sys.argv[1] if len(sys.argv) > 1 else "upsampled"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
keep = {}
orig = R.PoseRefiner.refine


def spy(self, *a, **k):
    keep["ref"] = self
    return orig(self, *a, **k)


R.PoseRefiner.refine = spy
bench.refinement_loop(dev, iters=2, graph=False, mode=mode)
ref = keep["ref"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ref._iteration()
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue                                                          # leaves of the operator tree only
    if not any(k.device == 0 or True for k in ev.kernels):                # operators that launched something
        continue
    if not ev.kernels:
        continue
    where = next((s_ for s_ in ev.stack if "nefes_amd" in s_ or "bench.py" in s_), ev.stack[0] if ev.stack else "?")
    rows[(ev.name, where.replace(ROOT + "/", ""), ev.kernels[0].name[:60])] += 1
for (name, where, kern), n in sorted(rows.items(), key=lambda kv: kv[0][1]):
    print(f"{n:3d} x {name:24s} {where:70s} {kern}")
