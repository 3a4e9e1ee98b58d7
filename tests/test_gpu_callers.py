"""The reference's validation callers of the path on the GPU: render_test (rendering.py:320-414),
render_path_upsample (:416-457), render_test_upsample (:459-491) and render_path_with_feature (:521-640) of the drop-in
`models.rendering`, each against the same thing composed by hand from one `render()` per pose and the torch expressions the
reference applies around it (per-image colour transform, bicubic resize, FusionNet, 10-pixel crop, 8-bit conversion, file names).
The PNG writer (imageio, not in this image) is replaced by a recorder; DFNet's feature extractor -- `dm.DFM_pose_refine`, the
reference's own module, absent on the GPU box -- by a stand-in module with the two names the caller imports."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dropin():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "nefes_amd", "dropin")
    if p not in sys.path:
        sys.path.insert(0, p)
    import models.rendering as R
    import models.nerfh_nff as M
    return R, M


def nets(Wd=128, C=16):
    from nefes_amd.field import NeRFH_NFF
    coarse = NeRFH_NFF('coarse', W=Wd, f_dim=C).requires_grad_(False).to(DEV)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    return coarse, fine


def setup(tmp_path, monkeypatch, **more):
    R, M = dropin()
    coarse, fine = nets()
    fields = dict(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1 << 21, encode_hist=True,
                  basedir=str(tmp_path), expname="exp", chunk=32768, render_test=False, render_video_train=False,
                  render_video_test=False, color_feat_loss=False, color_feat_fusion_loss=False, color_feat_fusion_nerfw_loss=False,
                  tinyscale=2)
    fields.update(more)
    args = types.SimpleNamespace(**fields)
    kw = dict(network_query_fn=None, perturb=False, N_importance=64, N_samples=64, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=True, args=args, ndc=False, lindisp=False, near=0., far=4.)
    written = {}
    monkeypatch.setattr(R, "_imwrite", lambda path, arr: written.__setitem__(os.path.relpath(path, str(tmp_path)), np.array(arr)))
    return R, M, coarse, fine, args, kw, written


def poses3():
    return torch.stack([O.bench_pose(), O.se3_exp_pose((0.2, 0.1, -0.1), (0.3, -0.2, 0.1)),
                        O.se3_exp_pose((-0.1, 0.3, 0.2), (-0.2, 0.1, 0.4))])


def hists(n):
    return (torch.arange(n * 10).reshape(n, 10) % 7 + 3).float()


def loader(images, poses, hist, batch):
    """what the reference's data loaders yield: dict batches (render_test) of img [b,3,H,W], pose [b,12], hist [b,10]"""
    n = images.shape[0]
    return [dict(img=images[i:i + batch], pose=poses[i:i + batch, :3, :4].reshape(-1, 12), hist=hist[i:i + batch])
            for i in range(0, n, batch)]


def one_by_one(R, H, W, focal, poses, kw):
    out = []
    with torch.no_grad():
        for b in range(poses.shape[0]):
            rgb, disp, acc, ex = R.render(H, W, focal, c2w=poses[b, :3, :4].to(DEV), **kw)
            out.append((rgb, disp, ex["feat_map"]))
    return out


def test_render_test_walks_both_loaders_and_writes_the_reference_files(tmp_path, monkeypatch):
    R, M, coarse, fine, args, kw, written = setup(tmp_path, monkeypatch)
    H, W, focal = 6, 8, 6.6
    poses, hist = poses3(), hists(3)
    p44 = torch.eye(4).repeat(3, 1, 1)
    p44[:, :3, :4] = poses
    g = torch.Generator().manual_seed(3)
    images = torch.rand(3, 3, H, W, generator=g)
    R.render_test(args, loader(images, p44, hist, 2), loader(images[:2], p44[:2], hist[:2], 1), (H, W, focal), 123, kw)
    assert torch.empty(1).device.type == "cpu"                                   # the default device is handed back (:366, :407)
    solo = one_by_one(R, H, W, focal, poses, kw)
    for split, n in (("train", 3), ("val", 2)):
        d = f"exp/evaluate_{split}_path_000123"
        assert os.path.isdir(os.path.join(str(tmp_path), d))
        assert sorted(k for k in written if k.startswith(d)) == sorted(f"{d}/{i:03d}{s}.png" for i in range(n) for s in ("", "_GT", "_disp"))
        for i in range(n):
            rgb, disp, _ = solo[i]
            rgb = coarse.affine_color_transform(args, rgb, hist[i:i + 1].to(DEV), 1)                 # rendering.py:276-278
            np.testing.assert_array_equal(written[f"{d}/{i:03d}.png"], R.to8b(rgb.reshape(H, W, 3).cpu().numpy()))
            dd = disp.reshape(H, W).cpu().numpy()
            np.testing.assert_array_equal(written[f"{d}/{i:03d}_disp.png"], R.to8b(dd / np.max(dd)))
            np.testing.assert_array_equal(written[f"{d}/{i:03d}_GT.png"], R.to8b(images[i].permute(1, 2, 0).numpy()))


def test_render_path_upsample_and_render_test_upsample(tmp_path, monkeypatch):
    R, M, coarse, fine, args, kw, written = setup(tmp_path, monkeypatch)
    H, W, focal = 6, 8, 6.6
    poses = poses3()
    solo = one_by_one(R, H, W, focal, poses, kw)
    with torch.no_grad():
        rgbs, disps = R.render_path_upsample(args, poses.to(DEV), (H, W, focal), 32768, kw, target_size=[20, 14])
    assert rgbs.shape == (3, 14, 20, 3) and disps == []
    for i in range(3):
        up = F.interpolate(solo[i][0].reshape(H, W, 3).permute(2, 0, 1)[None], size=(14, 20), mode='bicubic')[0].permute(1, 2, 0)
        np.testing.assert_array_equal(rgbs[i], up.cpu().numpy())
    with torch.no_grad():                                                             # target size = render size: no resize (:441)
        same, _ = R.render_path_upsample(args, poses.to(DEV), (H, W, focal), 32768, kw, target_size=[W, H])
    np.testing.assert_array_equal(same[1], solo[1][0].reshape(H, W, 3).cpu().numpy())
    # render_test_upsample: tuple batches of ONE image (img [1,3,H,W], pose [1,12], hist [1,10]) -> testset_renders/frame%05d.png
    p12 = poses.reshape(3, 12)
    dl = [(torch.zeros(1, 3, H, W), p12[i:i + 1], hists(3)[i:i + 1]) for i in range(3)]
    R.render_test_upsample(args, dl, (H, W, focal), kw, target_size=[20, 14])
    assert torch.empty(1).device.type == "cpu"
    d = "exp/testset_renders"
    assert sorted(k for k in written if k.startswith(d)) == [f"{d}/frame{i + 1:05d}.png" for i in range(3)]
    for i in range(3):
        np.testing.assert_array_equal(written[f"{d}/frame{i + 1:05d}.png"], R.to8b(rgbs[i]))


def test_render_path_with_feature_fuses_upsamples_and_crops(tmp_path, monkeypatch, capsys):
    R, M, coarse, fine, args, kw, written = setup(tmp_path, monkeypatch, color_feat_fusion_nerfw_loss=True)
    from nefes_amd.refine import feature_loss
    H, W, focal, ts = 32, 40, 33.0, 2
    h, w = H // ts, W // ts
    C = coarse.W_features
    poses, hist = poses3()[:2], hists(2).to(DEV)
    g = torch.Generator().manual_seed(4)
    gt_imgs = torch.rand(2, H, W, 3, generator=g).numpy()
    wf = torch.randn(C, 3, generator=g).to(DEV)

    def extractor(args_, target, device, feat_model, retFeature=True, isSingleStream=True, return_pose=False, H=None, W=None):
        assert tuple(target.shape) == (1, 3, H, W) and feat_model == "dfnet"
        feat = torch.einsum("cj,bjhw->bchw", wf, target)                     # any fixed function of the query image
        return [[feat]], None

    stub_pkg, stub = types.ModuleType("dm"), types.ModuleType("dm.DFM_pose_refine")
    stub.inference_pose_feature_extraction, stub.feature_loss = extractor, feature_loss
    stub_pkg.DFM_pose_refine = stub
    monkeypatch.setitem(sys.modules, "dm", stub_pkg)
    monkeypatch.setitem(sys.modules, "dm.DFM_pose_refine", stub)
    d = os.path.join(str(tmp_path), "out")
    os.makedirs(d)
    with torch.no_grad():
        out = R.render_path_with_feature(args, poses.to(DEV), (H, W, focal), 32768, kw, gt_imgs=gt_imgs, savedir=d, img_ids=hist,
                                         feat_model="dfnet", global_step=200)
    assert out == (None, None)                                                        # (:640: the reference returns nothing either)
    text = capsys.readouterr().out
    assert "Mean PSNR of this run is:" in text and "Feature cosine similarity loss:" in text
    assert sorted(written) == sorted(f"out/{i:03d}{s}.png" for i in range(2) for s in ("", "_GT", "_disp", "_feature_gt", "_feature"))
    solo = one_by_one(R, h, w, focal / ts, poses, kw)
    psnr, floss = [], []
    for i in range(2):
        rgb, disp, feat = solo[i]
        rgb = coarse.affine_color_transform(args, rgb, hist[i:i + 1], 1)
        render_rgb, _, feats = coarse.run_fusion_net(rgb, feat, h, w, B=1)
        up = torch.nn.Upsample(size=(H, W), mode='bicubic')
        rgb_up, feat_up = up(render_rgb), up(feats)[:, :, 10:-10, 10:-10]
        np.testing.assert_array_equal(written[f"out/{i:03d}.png"], R.to8b(rgb_up[0].permute(1, 2, 0).cpu().numpy()))
        target = torch.as_tensor(gt_imgs[i], device=DEV).permute(2, 0, 1)[None]
        gt_feat = torch.einsum("cj,bjhw->bchw", wf, target)[:, :, 10:-10, 10:-10]
        f0 = feat_up[0, 0]
        f0 = (f0 - f0.min()) / (f0.max() - f0.min())
        np.testing.assert_array_equal(written[f"out/{i:03d}_feature.png"], R.to8b(f0.cpu().numpy()))
        psnr.append(-10. * np.log10(np.mean(np.square((rgb_up - target).cpu().numpy()))))
        floss.append(float(feature_loss(feat_up[0], gt_feat[0], img_in=True, per_pixel=True)))
    said = {ln.split(":")[0]: float(ln.split(":")[1]) for ln in text.splitlines() if ln.count(":") == 1 and ln[0] in "MF"}
    assert abs(said["Mean PSNR of this run is"] - np.mean(psnr)) < 1e-4 and abs(said["Feature cosine similarity loss"] - np.mean(floss)) < 1e-5
    # before step 200 the maps go to the up-sampling without the FusionNet (:571-578)
    written.clear()
    with torch.no_grad():
        R.render_path_with_feature(args, poses[:1].to(DEV), (H, W, focal), 32768, kw, gt_imgs=gt_imgs, savedir=d, img_ids=hist,
                                   feat_model="dfnet", global_step=199)
    rgb = coarse.affine_color_transform(args, solo[0][0], hist[:1], 1).reshape(h, w, 3).permute(2, 0, 1)[None]
    np.testing.assert_array_equal(written["out/000.png"],
                                  R.to8b(torch.nn.Upsample(size=(H, W), mode='bicubic')(rgb)[0].permute(1, 2, 0).cpu().numpy()))
