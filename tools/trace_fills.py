"""Where do the small fill / copy launches of one refinement iteration come from?  (torch.profiler, eager PoseRefiner)"""
import collections, sys, types
import torch
sys.path.insert(0, ".")
from nefes_amd.field import NeRFH_NFF
from nefes_amd.refine import PoseRefiner
dev = "cuda"
C = 128
coarse = NeRFH_NFF('coarse', W=128, f_dim=C).requires_grad_(False).to(dev)
fine = NeRFH_NFF('fine', W=128, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(dev)
args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, encode_hist=True)
kw = dict(network_query_fn=None, perturb=0., N_importance=64, N_samples=64, network_fn=coarse, network_fine=fine, use_viewdirs=True,
          white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False)
up = "--no-upsample" not in sys.argv
r = PoseRefiner(kw, args, (240, 320, 262.75), 0., 4., tinyscale=4, upsample=up, graph=False, device=dev)
init = torch.eye(4, device=dev); hist = torch.full((1, 10), 10., device=dev)
target = torch.randn(C, 220, 300, device=dev) if up else torch.randn(C, 60, 80, device=dev)
r.refine(init, target, hist, 3)
from torch.profiler import profile, ProfilerActivity
from nefes_amd.refine import fix_coord_supp, feature_loss
from nefes_amd.render import render
from nefes_amd import ops


def census(tag, fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    n = t = 0
    names = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            n += 1; t += e.device_time; names[e.name[:48]] += 1
    print(f"{tag:34s} {n:4d} launches {t:8.1f} us   " + ", ".join(f"{k}:{v}" for k, v in names.most_common(4)))
    if "--ops" in sys.argv:
        cpu = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and (e.name.startswith("aten::") or "Backward" in e.name) and "evaluate_function" not in e.name)
        print("      ops: " + ", ".join(f"{k.replace('aten::','')}:{v}" for k, v in cpu.most_common(45)))


ws = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0., 0., 0.])
def pose_chain():
    c2w = fix_coord_supp(r.model(0)[None, :3, :4], ws)
    c2w.sum().backward()
census("pose chain fwd+bwd", pose_chain)
c2w0 = r.model(0)[:3, :4].detach().clone().requires_grad_()
def render_only():
    rgb, _, _, ex = render(r.h, r.w, r.focal, c2w=c2w0, near=0., far=4., img_idx=hist, **kw)
    (rgb.sum() + ex["feat_map"].sum()).backward()
census("render fwd+bwd (to c2w)", render_only)
rgb0 = torch.rand(r.h * r.w, 3, device=dev, requires_grad=True); feat0 = torch.randn(r.h * r.w, C, device=dev, requires_grad=True)
def fusion():
    x = coarse.affine_color_transform(args, rgb0, hist, 1)
    _, _, fused = coarse.run_fusion_net(x, feat0, r.h, r.w, 1)
    fused.sum().backward()
census("affine + fusion fwd+bwd", fusion)
fused0 = torch.randn(1, C, r.h, r.w, device=dev, requires_grad=True)
def tail():
    f = ops.bicubic_upsample(fused0, (r.H, r.W))[:, :, 10:-10, 10:-10] if up else fused0
    feature_loss(f[0], r.target).backward()
census("upsample + crop + loss fwd+bwd", tail)
def step():
    r.opt.step()
census("Adam step", step)
census("whole iteration", r._iteration)
