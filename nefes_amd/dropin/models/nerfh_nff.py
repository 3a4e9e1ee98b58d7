"""Drop-in for the reference's `models.nerfh_nff` (script/models/nerfh_nff.py): put `nefes_amd/dropin`
ahead of the reference's `script/` on sys.path and `import models.nerfh_nff` resolves here.
No tinycudann import; the field runs on the fused HIP kernels (nefes_amd.field / nefes_amd.ops)."""
import os

import numpy as np
import torch

from nefes_amd import lib as _L
from nefes_amd import ops as _ops
from nefes_amd.field import (FEATURE_DIM, ExposureMLP, FusionNet, NeRFH_NFF, get_embedder,  # noqa: F401
                             run_network_NeRFH_NFF)

APPLY_HISTOGRAM = True
img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def raw2outputs_NeRFH_NFF(raw, z_vals, raw_noise_std=0, output_transient=False, beta_min=0.1, white_bkgd=False,
                          test_time=False, typ="coarse", store_rgb=False, transient_at_test=False):
    """Reference signature (nerfh_nff.py:25).  raw [N,S,R] (any strides) -> the reference's 8-tuple."""
    dev = "cuda"
    raw = raw.to(dev)
    sigma_only = typ == "coarse" and test_time and not store_rgb
    if sigma_only:
        raw = raw[..., :1]
    raw_t = raw.permute(0, 2, 1).contiguous()
    C = raw_t.shape[1] - (1 if sigma_only else (6 if output_transient else 1)) - (0 if sigma_only else 3)
    flags = _L.COMP_SIGMA_ONLY if sigma_only else 0
    if not sigma_only and output_transient:
        flags |= _L.COMP_TRANSIENT
        if test_time and not transient_at_test:
            flags |= _L.COMP_STATIC_ONLY
        if white_bkgd:
            flags |= _L.COMP_WHITE_BKGD
    if raw_noise_std > 0 and not output_transient:
        noise = torch.zeros_like(raw_t)
        noise[:, 0 if sigma_only else 3 + C] = torch.randn(raw_t.shape[0], raw_t.shape[2], device=dev) * raw_noise_std
        raw_t = raw_t + noise
    rgb, feat, disp, acc, depth, weights, beta = _ops.Composite.apply(raw_t, z_vals.to(dev), max(C, 0), flags, float(beta_min))
    t_sig = raw_t[:, 3 + C + 4, :] if (output_transient and not sigma_only) else None
    if sigma_only:
        return None, None, None, acc, weights, None, t_sig, None
    return rgb, feat, disp, acc, weights, depth, t_sig, beta


def create_nerf(args):
    """nerfh_nff.py:628-736: same return tuple and render_kwargs keys; checkpoints load unchanged."""
    embed_fn, input_ch, _ = get_embedder(args.multires, args.i_embed, getattr(args, "reduce_embedding", -1))
    embeddirs_fn, input_ch_views = None, 0
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views, _ = get_embedder(args.multires_views, args.i_embed, getattr(args, "reduce_embedding", -1))
    device = torch.device("cuda")
    model = NeRFH_NFF('coarse', D=args.netdepth, W=args.netwidth, skips=[4], in_channels_xyz=input_ch,
                      in_channels_dir=input_ch_views, fusion_residule=getattr(args, "use_fusion_res", False),
                      no_BN=getattr(args, "no_fusion_BN", False)).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRFH_NFF('fine', D=args.netdepth, W=args.netwidth, skips=[4], in_channels_xyz=input_ch,
                               in_channels_dir=input_ch_views, encode_appearance=True, encode_transient=True,
                               in_channels_a=getattr(args, "in_channels_a", 50),
                               in_channels_t=getattr(args, "in_channels_t", 20)).to(device)
        grad_vars += list(model_fine.parameters())
    network_query_fn = lambda inputs, viewdirs, ts, network_fn, typ, output_transient, test_time, store_rgb: \
        run_network_NeRFH_NFF(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, typ=typ,
                              output_transient=output_transient, netchunk=args.netchunk, test_time=test_time,
                              store_rgb=store_rgb)
    if getattr(args, "no_grad_update", False):
        grad_vars, optimizer = None, None
    else:
        optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    if getattr(args, "ft_path", None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not getattr(args, "no_reload", False):
        print('Reloading from', ckpts[-1])
        ckpt = torch.load(ckpts[-1], map_location=device)
        start = ckpt['global_step']
        model.load_state_dict(ckpt['network_fn_state_dict'], strict=False)
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    render_kwargs_train = {'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
                           'N_samples': args.N_samples, 'network_fn': model, 'use_viewdirs': args.use_viewdirs,
                           'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std, 'test_time': False,
                           'args': args}
    if model_fine is not None:
        render_kwargs_train['network_fine'] = model_fine
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test.update(perturb=False, raw_noise_std=0., test_time=True)
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
