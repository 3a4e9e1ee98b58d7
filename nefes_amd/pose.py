"""6-DoF pose parameterisation of the refinement loop without lietorch.

Mirrors script/models/poses.py:6-50 (`LearnPose`) and script/utils/lie_group_helper.py:60-81 (`Exp`, `make_c2w`).
Twelve numbers per camera: plain torch, differentiable, no kernel needed.  `lietorch.SE3.exp` (poses.py:32,44) is a
CUDA extension the reference does not vendor; `se3_exp` below is the closed-form SE(3) exponential for lietorch's
[tau(3), phi(3)] tangent ordering (translation part V(phi)*tau), checked in tests against torch.linalg.matrix_exp of
the 4x4 twist.  Parity with lietorch itself is unpinned.
"""
import torch
import torch.nn as nn


_CONST = {}


def _consts(like):
    """Per (device, dtype): the so(3) generators G[i,j,k] (skew(v) = G @ v), eye(3) and the [0,0,0,1] row.  Built once, outside
    any graph capture that follows the first call, so a refinement iteration issues no fill / host-to-device copy for them."""
    key = (like.device, like.dtype)
    c = _CONST.get(key)
    if c is None:
        G = torch.zeros(3, 3, 3)
        G[0, 1, 2], G[0, 2, 1], G[1, 0, 2], G[1, 2, 0], G[2, 0, 1], G[2, 1, 0] = -1., 1., 1., -1., -1., 1.
        c = _CONST[key] = (G.to(like.device, like.dtype), torch.eye(3, device=like.device, dtype=like.dtype),
                           torch.tensor([[0., 0., 0., 1.]], device=like.device, dtype=like.dtype))
    return c


def _skew(v):
    """[[0,-z,y],[z,0,-x],[-y,x,0]] (lie_group_helper.vec2skew :45-57) as one contraction with the constant generators."""
    return torch.matmul(_consts(v)[0], v[..., None, :, None])[..., 0] if v.dim() > 1 else _consts(v)[0] @ v


def _bottom_row(like, shape):
    return _consts(like)[2].expand(*shape, 1, 4)


def so3_exp(r):
    """Rodrigues, same expression as lie_group_helper.Exp (:60-69): eps only guards the division."""
    K = _skew(r)
    n = r.norm(dim=-1, keepdim=True)[..., None] + 1e-15
    eye = _consts(r)[1].expand(K.shape)
    return eye + (torch.sin(n) / n) * K + ((1 - torch.cos(n)) / n ** 2) * (K @ K)


def make_c2w(r, t):
    """lie_group_helper.make_c2w (:72-81): [Exp(r) | t] as 4x4 (batched over leading dims)."""
    R = so3_exp(r)
    top = torch.cat([R, t[..., None]], -1)
    bottom = _bottom_row(r, top.shape[:-2])
    return torch.cat([top, bottom], -2)


def se3_exp(tau_phi):
    """SE(3) exponential of [tau, phi] (lietorch ordering) -> 4x4: R = Exp(phi), t = V(phi) tau,
    V = I + (1-cos th)/th^2 K + (th - sin th)/th^3 K^2 (series below th = 1e-4)."""
    tau, phi = tau_phi[..., :3], tau_phi[..., 3:]
    K = _skew(phi)
    th2 = (phi * phi).sum(-1)[..., None, None]
    th = torch.sqrt(th2.clamp_min(1e-30))
    small = th2 < 1e-8
    a = torch.where(small, 1 - th2 / 6, torch.sin(th) / th)
    b = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(th)) / th2.clamp_min(1e-30))
    c = torch.where(small, 1. / 6 - th2 / 120, (th - torch.sin(th)) / (th2 * th).clamp_min(1e-30))
    eye = _consts(phi)[1].expand(K.shape)
    R = eye + a * K + b * (K @ K)
    V = eye + b * K + c * (K @ K)
    t = (V @ tau[..., None])[..., 0]
    top = torch.cat([R, t[..., None]], -1)
    bottom = _bottom_row(phi, top.shape[:-2])
    return torch.cat([top, bottom], -2)


class LearnPose(nn.Module):
    """Same constructor, parameters (`r`, `t`, `init_c2w`) and forward semantics as the reference's LearnPose."""

    def __init__(self, num_cams, learn_R, learn_t, init_c2w=None, lietorch=False):
        super().__init__()
        self.num_cams, self.lietorch = num_cams, lietorch
        self.init_c2w = None if init_c2w is None else nn.Parameter(init_c2w, requires_grad=False)
        self.r = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_R)
        self.t = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_t)

    def forward(self, cam_id):
        r, t = self.r[cam_id], self.t[cam_id]
        c2w = se3_exp(torch.cat([t, r], -1)) if self.lietorch else make_c2w(r, t)
        if self.init_c2w is not None:                              # delta on top of the initial pose (poses.py:37-39,47-49)
            init = self.init_c2w[cam_id]
            R = c2w[..., :3, :3] @ init[..., :3, :3]
            tt = c2w[..., :3, 3] + init[..., :3, 3]
            c2w = torch.cat([torch.cat([R, tt[..., None]], -1), c2w[..., 3:, :]], -2)
        return c2w
