#!/usr/bin/env python3
"""Two query images refined at the same time, each as its own replayed graph on its own HIP stream: the latency-bound part of one
image's iteration (FusionNet's convolutions, the loss, the pose chain: ~0.3 ms of 1.7, a fraction of the chip) runs under the other
image's field kernels (power-bound, whole chip).  Prints ms per image for one stream, two streams, and eight images batched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nefes_amd import refine as R

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
made = []
orig = R.PoseRefiner.refine


def spy(self, *a, **k):
    if not any(m[0] is self for m in made):
        made.append((self, a, k))
    return orig(self, *a, **k)


R.PoseRefiner.refine = spy
n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(n_ref):
    sec1, _, _ = bench.refinement_loop(dev, iters=50, graph=True)
R.PoseRefiner.refine = orig
refs = [m[0] for m in made]
args = [m[1] for m in made]
streams = [torch.cuda.Stream(device=dev) for _ in refs]
iters, n_img = 50, 3


def run(active):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_img):
        for k in active:
            with torch.cuda.stream(streams[k]):
                refs[k]._reset(*[a.to(dev) for a in args[k][:3]])
        for i in range(iters):
            for k in active:
                with torch.cuda.stream(streams[k]):
                    refs[k].graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n_img * len(active))


run(list(range(len(refs))))
one = run([0])
many = run(list(range(len(refs))))
print(f"one image at a time: {one * 1e3:.2f} ms per image (bench line: {sec1 * 1e3:.2f}); {len(refs)} images on {len(refs)} streams: {many * 1e3:.2f} ms per image")
poses = []
for k in range(len(refs)):
    with torch.no_grad():
        poses.append(refs[k].model(0).detach().cpu())
print("final poses equal across streams:", all(torch.equal(poses[0], p) for p in poses[1:]))


def final(k):
    with torch.no_grad():
        return refs[k].model(0).detach().cpu()


run([0]); a = final(0)
run([0]); b = final(0)
run([1]); c = final(1)
print("same refiner twice, alone:", torch.equal(a, b), "| the other refiner, alone:", torch.equal(a, c), float((a - c).abs().max()))
many = run(list(range(len(refs))))
print("together: refiner 0 vs its own solo run:", torch.equal(final(0), a), float((final(0) - a).abs().max()), "| refiner 1 vs its solo run:",
      torch.equal(final(1), c), float((final(1) - c).abs().max()))
