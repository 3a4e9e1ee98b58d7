"""Every GPU launch of ONE refinement iteration (bench.py's loop50 set-up, eager PoseRefiner), in issue order, with the aten
operator that issued it: what is left to fuse.  torch.profiler."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nefes_amd.field import NeRFH_NFF
from nefes_amd.refine import PoseRefiner
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
wl = bench.WORKLOADS["ref"]
H, W, focal, Wd, C = wl["H"], wl["W"], wl["focal"], wl["Wd"], wl["C"]
coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(dev)
fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21, encode_hist=True)
kw = dict(network_query_fn=None, perturb=False, N_importance=wl["Ni"], N_samples=wl["Nc"], network_fn=coarse, network_fine=fine,
          use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
init = torch.eye(4, device=dev); init[:3, :4] = bench.bench_pose().to(dev)
hist = torch.full((1, 10), 10., device=dev)
target = torch.nn.functional.normalize(torch.randn(C, 4 * H - 20, 4 * W - 20, device=dev), dim=0)
r = PoseRefiner(kw, args, (4 * H, 4 * W, 4 * focal), 0., 4., tinyscale=4, upsample=True, graph=False, device=dev)
r.refine(init, target, hist, 3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r._iteration(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
tot = 0
for i, e in enumerate(ev):
    tot += e.device_time
    print(f"{i:3d} {e.device_time:8.1f} us  {e.name[:110]}")
print(f"{len(ev)} launches, {tot:.1f} us of kernel time")
