"""Drop-in for the reference's `models.rendering` (script/models/rendering.py): same public names.
render()/render_rays()/batchify_rays()/sample_pdf() run on the HIP kernels (nefes_amd.render);
render_path() keeps the reference's validation-loop call surface (rendering.py:246-318)."""
import os

import numpy as np
import torch

from models.nerfh_nff import mse2psnr, img2mse, raw2outputs_NeRFH_NFF, to8b  # noqa: F401
from models.ray_utils import get_rays, ndc_rays  # noqa: F401
from nefes_amd.render import batchify_rays, render, render_rays, sample_pdf  # noqa: F401

PROFILE_TIME = False
device = torch.device("cuda")


def render_path(args, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                single_gt_img=False, img_ids=torch.Tensor(0)):
    """rendering.py:246: loop render() over poses; returns (rgbs [n,H,W,3] np, disps [n,H,W] np), prints mean PSNR."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps, psnr = [], [], []
    for i, c2w in enumerate(render_poses):
        hist = img_ids[i:i + 1] if (torch.is_tensor(img_ids) and img_ids.numel() > 0) else torch.Tensor(0)
        with torch.no_grad():
            rgb, disp, acc, _ = render(int(H), int(W), focal, chunk=chunk, c2w=c2w[:3, :4], img_idx=hist, **render_kwargs)
        rgb, disp = rgb.reshape(int(H), int(W), 3), disp.reshape(int(H), int(W))
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if gt_imgs is not None and render_factor == 0:
            gt = gt_imgs if single_gt_img else gt_imgs[i]
            psnr.append(float(mse2psnr(img2mse(rgb, torch.as_tensor(gt, device=rgb.device, dtype=rgb.dtype)))))
        if savedir is not None:
            import imageio   # optional dependency, only for PNG dumps
            imageio.imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
    if psnr:
        print("Mean PSNR of this run is:", float(np.mean(psnr)))
    return np.stack(rgbs, 0), np.stack(disps, 0)
