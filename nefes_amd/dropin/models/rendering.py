"""Drop-in for the reference's `models.rendering` (script/models/rendering.py): same public names and signatures.

render()/render_rays()/batchify_rays()/sample_pdf() run on the HIP kernels (nefes_amd.render).  The validation
helpers -- render_path (:246), render_test (:320), render_path_upsample (:416), render_test_upsample (:459),
render_path_with_feature (:521) -- keep the reference's call surface and outputs, but render their poses in batches
through nefes_amd.render.render_poses (one launch sequence for several poses) instead of one render() per pose.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from models.nerfh_nff import mse2psnr, img2mse, raw2outputs_NeRFH_NFF, to8b  # noqa: F401
from models.ray_utils import get_rays, ndc_rays  # noqa: F401
from nefes_amd.render import batchify_rays, render, render_rays, sample_pdf  # noqa: F401
from nefes_amd.render import render_poses as _render_poses

PROFILE_TIME = False
POSES_PER_LAUNCH = 8           # upper bound of poses rendered by one launch sequence (memory permitting)
device = torch.device("cuda")


def _imwrite(path, arr):
    import imageio                                       # optional dependency of the reference, only needed for PNG dumps
    imageio.imwrite(path, arr)


def _scaled(hwf, render_factor):
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = int(H // render_factor), int(W // render_factor), focal / render_factor
    return int(H), int(W), focal


def _hist(img_ids, i):
    """img_ids[i:i+1] as the reference indexes it (one histogram row per pose); an empty tensor means no histogram."""
    if torch.is_tensor(img_ids) and img_ids.numel() > 0:
        return img_ids[i:i + 1]
    return torch.Tensor(0)


def _render_group(H, W, focal, chunk, poses, render_kwargs):
    """rgb [B,H*W,3], disp [B,H*W], extras for a group of poses in one launch sequence."""
    kw = {k: v for k, v in render_kwargs.items() if k != "img_idx"}
    rgb, disp, acc, extras = _render_poses(H, W, focal, poses, chunk=chunk, **kw)
    return rgb, disp, extras


def _groups(n):
    for i0 in range(0, n, POSES_PER_LAUNCH):
        yield i0, min(n, i0 + POSES_PER_LAUNCH)


def _affine(args, render_kwargs, rgb, hist):
    """rendering.py:276-278: per-image affine colour transform when histograms are encoded."""
    if getattr(args, "encode_hist", False) and (getattr(args, "sh_nff", False) or getattr(args, "nerfh_nff", False)
                                                or getattr(args, "nerfh_nff2", False)):
        return render_kwargs['network_fn'].affine_color_transform(args, rgb, hist, 1)
    return rgb


def _gt_of(gt_imgs, i, single_gt_img):
    gt = gt_imgs if single_gt_img else gt_imgs[i]
    return gt.cpu().numpy() if torch.is_tensor(gt) else gt


def render_path(args, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                single_gt_img=False, img_ids=torch.Tensor(0)):
    """rendering.py:246-318: render every pose, colour-correct, PSNR against gt_imgs, optional PNG dumps
    ({i:03d}.png, _GT.png, _disp.png).  Returns (rgbs [n,H,W,3] np, disps [n,H,W] np)."""
    H, W, focal = _scaled(hwf, render_factor)
    rgbs, disps, psnr = [], [], []
    n = len(render_poses)
    for i0, i1 in _groups(n):
        poses = torch.stack([torch.as_tensor(render_poses[i])[:3, :4] for i in range(i0, i1)]).to(device)
        with torch.no_grad():
            rgb_g, disp_g, _ = _render_group(H, W, focal, chunk, poses, render_kwargs)
        for i in range(i0, i1):
            rgb = _affine(args, render_kwargs, rgb_g[i - i0], _hist(img_ids, i))
            rgbs.append(rgb.reshape(H, W, 3).cpu().numpy())
            disps.append(disp_g[i - i0].reshape(H, W).cpu().numpy())
            if i == 0:
                print(rgb.shape, disp_g[i - i0].shape)
            if gt_imgs is not None:
                gt = _gt_of(gt_imgs, i, single_gt_img)
                psnr.append(-10. * np.log10(np.mean(np.square(rgbs[i] - gt))))
            if savedir is not None:
                _imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
                if gt_imgs is not None:
                    _imwrite(os.path.join(savedir, '{:03d}_GT.png'.format(i)), to8b(_gt_of(gt_imgs, i, single_gt_img)))
                _imwrite(os.path.join(savedir, '{:03d}_disp.png'.format(i)), to8b(disps[-1] / np.max(disps[-1])))
    rgbs, disps = np.stack(rgbs, 0), np.stack(disps, 0)
    print("Mean PSNR of this run is:", np.mean(psnr, 0) if psnr else float("nan"))
    return rgbs, disps


def _collect(dl, batched):
    """Poses / images / histograms of a data loader as the reference gathers them (rendering.py:330-347, 470-480)."""
    images, poses, index = [], [], []
    for batch in dl:
        if batched:
            img, pose, hist = batch['img'], batch['pose'], batch['hist']
        else:
            img, pose, hist = batch
            img, pose, hist = img[:1], pose.reshape(1, -1), hist
        for i in range(img.shape[0]):
            p = torch.zeros(1, 4, 4)
            p[0, :3, :4] = pose[i:i + 1].reshape(3, 4)[:3, :4]
            p[0, 3, 3] = 1.
            images.append(img[i:i + 1].permute(0, 2, 3, 1))
            poses.append(p)
            index.append(hist[i:i + 1] if batched else hist)
    return torch.cat(images, 0).numpy(), torch.cat(poses, 0).to(device), torch.cat(index, 0).to(device)


def _with_feature(args):
    return bool(getattr(args, "color_feat_loss", False) or getattr(args, "color_feat_fusion_loss", False)
                or getattr(args, "color_feat_fusion_nerfw_loss", False))


def render_test(args, train_dl, val_dl, hwf, start, render_kwargs_test, feat_model=None, pose_param_net=None):
    """rendering.py:320-414: evaluate the training and the validation poses, dumps under
    basedir/expname/evaluate_{train,val}_{test|path}_{start:06d}."""
    tag = 'test' if getattr(args, "render_test", False) else 'path'
    for split, dl in (("train", train_dl), ("val", val_dl)):
        savedir = os.path.join(args.basedir, args.expname, 'evaluate_{}_{}_{:06d}'.format(split, tag, start))
        os.makedirs(savedir, exist_ok=True)
        images, poses, index = _collect(dl, batched=True)
        print('{} poses shape'.format('train' if split == "train" else 'test'), poses.shape)
        with torch.no_grad():
            torch.set_default_device('cuda')
            try:
                if _with_feature(args):
                    rgbs, disps = render_path_with_feature(args, poses, hwf, args.chunk, render_kwargs_test, gt_imgs=images,
                                                           savedir=savedir, img_ids=index, feat_model=feat_model,
                                                           global_step=start)
                else:
                    rgbs, disps = render_path(args, poses, hwf, args.chunk, render_kwargs_test, gt_imgs=images,
                                              savedir=savedir, img_ids=index)
            finally:
                torch.set_default_device('cpu')
        print('Saved {} set'.format('train' if split == "train" else 'test'))
        if getattr(args, "render_video_" + ("train" if split == "train" else "test"), False) and rgbs is not None:
            import imageio
            base = os.path.join(args.basedir, args.expname,
                                '{}_{}_{:06d}_'.format(args.expname, 'trainset' if split == "train" else 'test', start))
            name = 'train' if split == "train" else 'test'
            imageio.mimwrite(base + name + '_rgb.mp4', to8b(rgbs), fps=15, quality=8)
            imageio.mimwrite(base + name + '_disp.mp4', to8b(disps / np.max(disps)), fps=15, quality=8)
        del images, poses
        torch.cuda.empty_cache()


def render_path_upsample(args, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                         single_gt_img=False, img_ids=torch.Tensor(0), target_size=[-1, -1]):
    """rendering.py:416-457: render at hwf, bicubic-resize the colour image to target_size = (W_t, H_t), dump
    frame{i+1:05d}.png.  Returns (rgbs [n,H_t,W_t,3] np, [])."""
    H, W, focal = _scaled(hwf, render_factor)
    rgbs, disps = [], []
    for i0, i1 in _groups(len(render_poses)):
        poses = torch.stack([torch.as_tensor(render_poses[i])[:3, :4] for i in range(i0, i1)]).to(device)
        rgb_g, disp_g, _ = _render_group(H, W, focal, chunk, poses, render_kwargs)
        for i in range(i0, i1):
            rgb = rgb_g[i - i0].reshape(H, W, 3)
            if target_size[0] != W or target_size[1] != H:
                rgb = F.interpolate(rgb.permute(2, 0, 1)[None], size=(target_size[1], target_size[0]), mode='bicubic')
                rgb = rgb[0].permute(1, 2, 0)
            rgbs.append(rgb.cpu().numpy())
            if i == 0:
                print(rgb.shape, disp_g[i - i0].shape)
            if savedir is not None:
                _imwrite(os.path.join(savedir, 'frame{:05d}.png'.format(i + 1)), to8b(rgbs[-1]))
    return np.stack(rgbs, 0), disps


def render_test_upsample(args, val_dl, hwf, render_kwargs_test, target_size=[-1, -1]):
    """rendering.py:459-491: validation poses rendered at hwf and up-sampled to target_size, dumps under
    basedir/expname/testset_renders."""
    savedir = os.path.join(args.basedir, args.expname, 'testset_renders')
    os.makedirs(savedir, exist_ok=True)
    images, poses, index = _collect(val_dl, batched=False)
    print('test poses shape', poses.shape)
    with torch.no_grad():
        torch.set_default_device('cuda')
        try:
            render_path_upsample(args, poses, hwf, args.chunk, render_kwargs_test, gt_imgs=images, savedir=savedir,
                                 img_ids=index, target_size=target_size)
        finally:
            torch.set_default_device('cpu')
    print('Saved test set')


def _saliency_png(path, feat):
    """First channel of a [1,C,H,W] feature map as a min-max normalised grey PNG (what plot_features :507-519 dumps through
    utils.save_image_saliancy)."""
    f = feat[0, 0].detach().float()
    f = (f - f.min()) / (f.max() - f.min()).clamp_min(1e-12)
    _imwrite(path, to8b(f.cpu().numpy()))


def render_path_with_feature(args, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                             single_gt_img=False, img_ids=torch.Tensor(0), feat_model=None, global_step=None):
    """rendering.py:521-640: render at 1/tinyscale, colour-correct, (global_step >= 200) fuse rgb+features with the
    FusionNet, bicubic-upsample features and colour to (H, W), crop 10 px, report the mean colour PSNR and the mean
    feature cosine loss against features extracted from gt_imgs by `feat_model`.  Returns (None, None) like the
    reference."""
    from dm.DFM_pose_refine import inference_pose_feature_extraction, feature_loss   # reference side (DFNet plumbing)
    assert feat_model is not None
    H, W, focal = _scaled(hwf, render_factor)
    ts = args.tinyscale
    h, w = int(H // ts), int(W // ts)
    net = render_kwargs['network_fn']
    psnr, feats_psnr = [], []
    for i0, i1 in _groups(len(render_poses)):
        poses = torch.stack([torch.as_tensor(render_poses[i])[:3, :4] for i in range(i0, i1)]).to(device)
        rgb_g, disp_g, extras = _render_group(h, w, focal / ts, chunk, poses, render_kwargs)
        for i in range(i0, i1):
            rgb = rgb_g[i - i0]
            if getattr(args, "encode_hist", False):
                rgb = net.affine_color_transform(args, rgb, _hist(img_ids, i), 1)
            feat = extras['feat_map'][i - i0]
            if global_step >= 200:
                render_rgb, _, feats = net.run_fusion_net(rgb, feat, h, w, B=1)                  # [1,3,h,w], [1,C,h,w]
            else:
                render_rgb = rgb.reshape(h, w, 3).permute(2, 0, 1)[None]
                feats = feat.reshape(h, w, -1).permute(2, 0, 1)[None]
            feat_map = torch.nn.Upsample(size=(H, W), mode='bicubic')(feats)
            if i == 0:
                print(rgb.shape, disp_g[i - i0].shape)
            target = torch.as_tensor(gt_imgs[i], dtype=torch.float32, device=render_rgb.device).permute(2, 0, 1)[None]
            with torch.no_grad():
                gt_feat, _ = inference_pose_feature_extraction(args, target, device, feat_model, retFeature=True,
                                                               isSingleStream=True, return_pose=False, H=H, W=W)
                gt_feat = gt_feat[0][0].detach()
                render_rgb_up = torch.nn.Upsample(size=(H, W), mode='bicubic')(render_rgb)
            gt_feat, feat_map = gt_feat[:, :, 10:-10, 10:-10], feat_map[:, :, 10:-10, 10:-10]
            psnr.append(-10. * np.log10(np.mean(np.square((render_rgb_up - target).cpu().numpy()))))
            if savedir is not None:
                _imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(render_rgb_up[0].permute(1, 2, 0).cpu().numpy()))
                _imwrite(os.path.join(savedir, '{:03d}_GT.png'.format(i)), to8b(gt_imgs[i]))
                d = disp_g[i - i0].reshape(h, w).cpu().numpy()
                _imwrite(os.path.join(savedir, '{:03d}_disp.png'.format(i)), to8b(d / np.max(d)))
                _saliency_png(os.path.join(savedir, '{:03d}_feature_gt.png'.format(i)), gt_feat)
                _saliency_png(os.path.join(savedir, '{:03d}_feature.png'.format(i)), feat_map)
                feats_psnr.append(feature_loss(feat_map[0], gt_feat[0], img_in=True, per_pixel=True).cpu().numpy())
    print("Mean PSNR of this run is:", np.mean(psnr, 0))
    print("Feature cosine similarity loss:", np.mean(feats_psnr, 0) if feats_psnr else float("nan"))
    return None, None
