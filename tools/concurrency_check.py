import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from nefes_amd import refine as R
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
made = []
orig = R.PoseRefiner.refine
def spy(self, *a, **k):
    if not any(m[0] is self for m in made): made.append((self, a, k))
    return orig(self, *a, **k)
R.PoseRefiner.refine = spy
for _ in range(2): bench.refinement_loop(dev, iters=5, graph=False)
R.PoseRefiner.refine = orig
refs = [m[0] for m in made]; args = [m[1] for m in made]
streams = [torch.cuda.Stream(device=dev) for _ in refs]
from nefes_amd import ops
from nefes_amd.render import render
def stage(k, what):
    r = refs[k]
    c2w = r.model.init_c2w[0, :3, :4].clone().requires_grad_()
    rgb, _, _, ex = render(r.h, r.w, r.focal, c2w=c2w, near=r.near, far=r.far, img_idx=r.hist, **r.kw)
    feat = ex["feat_map"]
    if what == "render_fwd":
        return torch.cat([rgb.detach().reshape(-1), feat.detach().reshape(-1)])
    if what == "render_rgb_only":
        (g,) = torch.autograd.grad((rgb ** 2).sum(), c2w); return g
    if what == "render_plain_kernels":
        pass
    if what.startswith("render"):
        loss = (rgb ** 2).sum() + (feat ** 2).sum()
        (g,) = torch.autograd.grad(loss, c2w); return g
    fnet = r.coarse.fusion_net
    x = ops.fusion_input(rgb.detach(), feat.detach(), r._affine, 1, r.h, r.w, fnet.mean, fnet.std).requires_grad_()
    if what == "convs":
        y = fnet._convs_hip(x)
    else:
        y = fnet.forward_prepared(x)
    (g,) = torch.autograd.grad((y ** 2).sum(), x); return g
for k in range(2):
    refs[k]._reset(*[a.to(dev) for a in args[k][:3]])
for what in ("render_fwd", "render_rgb_only", "render", "render_plain_kernels", "convs", "fusion_net_with_bn"):
    ops.FACTORED_HEAD = what != "render_plain_kernels"
    solo = [stage(k, what).clone() for k in range(2)]
    torch.cuda.synchronize()
    bad = 0
    for rep in range(20):
        outs = []
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs.append(stage(k, what))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, s) else 1 for o, s in zip(outs, solo))
    print(what, "concurrent results different from solo:", bad, "of 40")
