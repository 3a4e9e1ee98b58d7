import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from nefes_amd import refine as R
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
keep = {}
orig = R.PoseRefiner.refine_apr
def spy(self, *a, **k):
    keep["ref"], keep["a"] = self, a
    return orig(self, *a, **k)
R.PoseRefiner.refine_apr = spy
bench.refinement_loop(dev, iters=50, graph=True, mode="2")
R.PoseRefiner.refine_apr = orig
ref = keep["ref"]; photo, full, hist = keep["a"][:3]
for ver in (True, False, True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ref.refine_apr(photo, full, hist, 50, verification=ver)
    torch.cuda.synchronize()
    print("verification", ver, round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per image", flush=True)
# per-image fixed cost: iterations = 0
for it in (0, 50, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ref.refine_apr(photo, full, hist, it, verification=False)
    torch.cuda.synchronize()
    print("iters", it, round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per image", flush=True)
# the same with the query image, its features and the histogram already on the device
photo_d, full_d, hist_d = photo.to(dev), full.to(dev), hist.to(dev)
for it in (0, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ref.refine_apr(photo_d, full_d, hist_d, it, verification=False)
    torch.cuda.synchronize()
    print("iters", it, "inputs resident", round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per image", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ref.refine_apr(photo_d, full_d, hist_d, 50, verification=True)
torch.cuda.synchronize()
print("iters 50 inputs resident, with verification", round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per image", flush=True)
