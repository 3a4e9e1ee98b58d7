#!/usr/bin/env python3
"""Generate tests/golden/refine.npz and tests/golden/affine.npz by running the REFERENCE's own refinement-loop functions
(/root/reference, CPU, this container only; the reference never travels -- these vectors do).

refine.npz -- BASELINE configs[4]/[5] in miniature: 12 iterations of the per-image loop of
script/dm/DFM_pose_refine.py:290-348 (`DFM_optimization_NFF`), executed with the reference's objects:

    models.poses.LearnPose (lietorch=False: utils/lie_group_helper.make_c2w)  ->  dm.direct_pose_model.fix_coord_supp
      -> models.rendering.render (12x16 rays = 48x64 / tinyscale 4, 16+16 samples, 8x128 MLP, C=128, test-time kwargs)
      -> NeRFH_NFF.affine_color_transform(hist)  ->  NeRFH_NFF.run_fusion_net (BatchNorm in train mode, as the reference runs it)
      -> dm.DFM_pose_refine.feature_loss (1 - mean cosine similarity)  ->  backward  ->  torch.optim.Adam(lr_r, lr_t)

`DFM_optimization_NFF` itself cannot be called here: it switches torch's default device to 'cuda' (:316) and this container has
no GPU.  The driver below makes the same calls in the same order (:310-341) on the CPU.  The target features are the same
pipeline's output at the ground-truth pose (the reference takes them from DFNet, a CNN outside the path).

affine.npz -- NeRFH_NFF.affine_color_transform (nerfh_nff.py:605-626) on a batch of two histograms.

tcnn (`exposure_embedding`, nerfh_nff.py:511-522) is not installed and not vendored by the reference: `FlatMLP` below stands in
for `tcnn.Network` with the flat parameter vector laid out as nefes_amd.field.ExposureMLP reads it and fp32 arithmetic.  What
these fixtures pin is therefore the reference's code AROUND that network (histogram cast, 3x3 kernel / bias split, bmm, sigmoid,
fusion CNN, loss, pose chain, optimizer); tiny-cuda-nn's own layout and fp16 arithmetic stay UNPINNED.

Usage:  python tools/make_golden_refine.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

# third-party modules the reference imports at module level and this container lacks; none is called on the path
MISSING = ("tinycudann", "imageio", "cv2", "torchvision", "lietorch", "torchsummary", "kornia", "efficientnet_pytorch",
           "pytorch3d", "tensorboardX", "wandb", "transforms3d", "pykalman")


class FlatMLP(torch.nn.Module):
    """Stand-in for tcnn.Network(10 -> 12, FullyFusedMLP, 32 neurons, 3 hidden layers, ReLU, no bias, no output activation)."""
    SHAPES = [(32, 16), (32, 32), (32, 32), (16, 32)]

    def __init__(self, n_input_dims, n_output_dims, network_config):
        super().__init__()
        assert (n_input_dims, n_output_dims) == (10, 12) and network_config["n_neurons"] == 32 and network_config["n_hidden_layers"] == 3
        self.params = torch.nn.Parameter(torch.zeros(sum(o * i for o, i in self.SHAPES)))

    def forward(self, x):
        h = torch.nn.functional.pad(x.float(), (0, 6))
        off = 0
        for k, (o, i) in enumerate(self.SHAPES):
            h = h @ self.params[off:off + o * i].view(o, i).t()
            off += o * i
            if k < 3:
                h = torch.relu(h)
        return h[:, :12]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        return importlib.machinery.ModuleSpec(name, self, is_package=True) if name.split(".")[0] in MISSING else None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        if spec.name == "tinycudann":
            m.Network = FlatMLP

        def _attr(k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Dummy()
        m.__getattr__ = _attr
        return m

    def exec_module(self, m):
        pass


def import_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path[:0] = [REF + "/script", REF]
    import dm.DFM_pose_refine as DR
    import models.nerfh_nff as M
    return DR, M


def npy(t):
    return t.detach().cpu().numpy()


def rot(axis, deg):
    a = torch.tensor(axis, dtype=torch.float64)
    a = a / a.norm()
    th = np.deg2rad(deg)
    K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=torch.float64)
    return (torch.eye(3, dtype=torch.float64) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)).float()


def main():
    os.makedirs(OUT, exist_ok=True)
    DR, M = import_reference()
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(4321)

    Wd, C, Nc, Ni = 128, 128, 16, 16      # the refinement shape of the reference (8x128 MLP, 128 feature channels)
    H, W, focal, ts = 48, 64, 60.0, 4
    near, far = 0.5, 4.0
    coarse = M.NeRFH_NFF('coarse', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27, f_dim=C)
    fine = M.NeRFH_NFF('fine', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True,
                       encode_transient=True, in_channels_a=50, in_channels_t=20, f_dim=C)
    expo = (torch.rand(coarse.exposure_embedding.params.numel(), generator=g) - 0.5) * 0.08
    with torch.no_grad():
        coarse.exposure_embedding.params.copy_(expo)
    coarse.requires_grad_(False)           # DFM_post_processing2 (:365-366)
    fine.requires_grad_(False)

    embed_fn, _, _ = M.get_embedder(10, 0, -1)
    embeddirs_fn, _, _ = M.get_embedder(4, 0, -1)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21,
                                 encode_hist=True, tinyscale=ts, chunk=1 << 15, lr_r=0.01, lr_t=0.01)
    q = lambda inputs, viewdirs, ts_, network_fn, typ, output_transient, test_time, store_rgb: \
        M.run_network_NeRFH_NFF(inputs, viewdirs, ts_, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, typ=typ,
                                output_transient=output_transient, netchunk=args.netchunk, test_time=test_time,
                                store_rgb=store_rgb)
    kw = dict(network_query_fn=q, perturb=0., N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False,
              near=near, far=far)
    world = {"pose_scale": 0.8, "pose_scale2": 1.25, "move_all_cam_vec": [0.1, -0.05, 0.2]}
    hist = torch.tensor([[3., 7., 12., 20., 31., 18., 9., 4., 2., 1.]])
    h, w = H // ts, W // ts

    def pipeline(pose_net, cam):
        """DFM_optimization_NFF (:310-334) on the CPU."""
        pose = pose_net(cam_id=cam)[None, :3, :4]
        pose_nerf = DR.fix_coord_supp(args, pose, world, device=None)
        rgb, _, _, extras = DR.render(h, w, focal / ts, chunk=args.chunk, c2w=pose_nerf[0, :3, :4], img_idx=hist, **kw)
        rgb = coarse.affine_color_transform(args, rgb, hist, 1)
        _, _, feature_rgb = coarse.run_fusion_net(rgb, extras['feat_map'], h, w, 1)
        return feature_rgb

    true_c2w = torch.eye(4)
    true_c2w[:3, :3] = rot((0.2, 1.0, -0.1), 8.0)
    true_c2w[:3, 3] = torch.tensor([0.15, -0.10, 0.30])
    init_c2w = torch.eye(4)
    init_c2w[:3, :3] = rot((1.0, 0.3, 0.5), 10.0) @ true_c2w[:3, :3]
    init_c2w[:3, 3] = true_c2w[:3, 3] + torch.tensor([0.30, -0.20, 0.25])

    with torch.no_grad():
        gt_net = DR.LearnPose(1, True, True, true_c2w[None].clone(), lietorch=False)
        target = pipeline(gt_net, 0).clone()                     # [1,C,h,w]

    pose_net = DR.LearnPose(1, True, True, init_c2w[None].clone(), lietorch=False)
    params = []
    for name, p in pose_net.named_parameters():                  # DFM_post_processing2 (:392-398)
        if name == 'r':
            params.append({'params': p, 'lr': args.lr_r})
        elif name == 't':
            params.append({'params': p, 'lr': args.lr_t})
    opt = torch.optim.Adam(params)
    n_iter = 12
    losses, poses, rs, tsv, grads = [], [], [], [], []
    pose_net.train()
    for it in range(n_iter):
        feature_rgb = pipeline(pose_net, 0)
        loss = DR.feature_loss(feature_rgb[0], target[0], per_pixel=False)
        loss.backward()
        grads.append(np.concatenate([npy(pose_net.r.grad[0]), npy(pose_net.t.grad[0])]))
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
        with torch.no_grad():
            poses.append(npy(pose_net(cam_id=0)[:3, :4]))
        rs.append(npy(pose_net.r[0]).copy())
        tsv.append(npy(pose_net.t[0]).copy())
        print(f"iter {it:2d}  loss {losses[-1]:.6f}  r {rs[-1]}  t {tsv[-1]}")
    bn = coarse.fusion_net.net[-1]
    np.savez_compressed(os.path.join(OUT, "refine.npz"), Wd=Wd, C=C, Nc=Nc, Ni=Ni, hwf=np.array([H, W, focal]), tinyscale=ts,
                        near=near, far=far, lr=np.array([args.lr_r, args.lr_t]), hist=npy(hist), exposure_params=npy(expo),
                        pose_scale=world["pose_scale"], pose_scale2=world["pose_scale2"],
                        move_all_cam_vec=np.array(world["move_all_cam_vec"]), true_c2w=npy(true_c2w), init_c2w=npy(init_c2w),
                        target=npy(target[0]), losses=np.array(losses), poses=np.stack(poses), r=np.stack(rs), t=np.stack(tsv),
                        grads=np.stack(grads), bn_running_mean=npy(bn.running_mean), bn_running_var=npy(bn.running_var),
                        bn_batches=int(bn.num_batches_tracked))

    # ---- affine_color_transform on its own (two images, 7 rays each) ----------------------------------------------------
    hist2 = torch.tensor([[3., 7., 12., 20., 31., 18., 9., 4., 2., 1.], [0., 0., 1., 5., 40., 60., 22., 3., 0., 0.]])
    rgb_in = torch.rand(14, 3, generator=g)
    rgb_out = coarse.affine_color_transform(args, rgb_in.clone(), hist2, 2)
    np.savez_compressed(os.path.join(OUT, "affine.npz"), exposure_params=npy(expo), hist=npy(hist2), rgb_in=npy(rgb_in),
                        rgb_out=npy(rgb_out), a_embedded=npy(coarse.a_embedded))
    print("wrote refine.npz, affine.npz")


if __name__ == "__main__":
    main()
