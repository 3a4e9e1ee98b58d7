"""Drop-in for the reference's `models.ray_utils` (script/models/ray_utils.py): same names, HIP kernels inside."""
import numpy as np
import torch

from nefes_amd import ops


def get_rays(H, W, focal, c2w):
    """ray_utils.py:5-16.  Returns rays_o, rays_d [H,W,3] on the GPU, differentiable w.r.t. c2w."""
    o, d, _ = ops.RayGen.apply(c2w.to("cuda"), int(H), int(W), float(focal), 0, int(H))
    return o.reshape(H, W, 3), d.reshape(H, W, 3)


def get_rays_batch(H, W, focal, c2w):
    """ray_utils.py:46-59: [B,3,4] poses -> [B,H,W,3] x2."""
    assert c2w.dim() == 3
    pairs = [get_rays(H, W, focal, c2w[k]) for k in range(c2w.shape[0])]
    return torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ray_utils.py:27-44."""
    return ops.NdcRays.apply(rays_o.to("cuda"), rays_d.to("cuda"), int(H), int(W), float(focal), float(near))


def get_rays_np(H, W, focal, c2w):
    """ray_utils.py:18-25 (numpy, host side; used by data loaders only)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    return np.broadcast_to(c2w[:3, -1], np.shape(rays_d)), rays_d
