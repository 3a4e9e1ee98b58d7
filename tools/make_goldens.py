#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference)
on CPU in the build container.  The reference never travels to the GPU box;
these vectors (inputs + expected outputs only) do.

Usage:  python tools/make_goldens.py            (writes tests/golden/)

The reference has no tests of its own, so these vectors are what
pins the oracle (`oracle/ref_cpu.py`) and, through it, the HIP path.
Import recipe: SURVEY.md Appendix C (four import-time stubs; the path never
calls into them).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Net(torch.nn.Module):      # stands in for tcnn.Network(10 -> 12); never called on the path
        def __init__(self, n_input_dims, n_output_dims, network_config):
            super().__init__()
            self.l = torch.nn.Linear(n_input_dims, n_output_dims, bias=False)

        def forward(self, x):
            return self.l(x.float())

    stub("tinycudann", Network=_Net, Encoding=None)
    stub("imageio")
    stub("cv2")
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", make_grid=None, save_image=None)
    sys.path[:0] = [REF + "/script", REF]
    import models.rendering as R
    import models.nerfh_nff as M
    import models.ray_utils as RU
    return R, M, RU


def npy(t):
    return t.detach().cpu().numpy()


def build_nets(M, Wd, C, sigma_scale=1.0):
    coarse = M.NeRFH_NFF('coarse', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27, f_dim=C)
    fine = M.NeRFH_NFF('fine', D=8, W=Wd, skips=[4], in_channels_xyz=63, in_channels_dir=27,
                       encode_appearance=True, encode_transient=True, in_channels_a=50, in_channels_t=20, f_dim=C)
    if sigma_scale != 1.0:
        with torch.no_grad():
            for net in (coarse, fine):
                net.static_sigma[0].weight.mul_(sigma_scale)
                net.static_sigma[0].bias.mul_(sigma_scale)
    return coarse, fine


def make_kwargs(M, coarse, fine, Nc, Ni, transient_at_test, test_time=True, perturb=0.):
    embed_fn, _, _ = M.get_embedder(10, 0, -1)
    embeddirs_fn, _, _ = M.get_embedder(4, 0, -1)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True,
                                 transient_at_test=transient_at_test, netchunk=1 << 21)
    q = lambda inputs, viewdirs, ts, network_fn, typ, output_transient, test_time, store_rgb: \
        M.run_network_NeRFH_NFF(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                typ=typ, output_transient=output_transient, netchunk=args.netchunk,
                                test_time=test_time, store_rgb=store_rgb)
    return dict(network_query_fn=q, perturb=perturb, N_importance=Ni, N_samples=Nc, network_fn=coarse,
                network_fine=fine, use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=test_time,
                args=args, ndc=False, lindisp=False)


def pose(r, t):
    sys.path.insert(0, ROOT)
    from oracle.ref_cpu import se3_exp_pose
    return se3_exp_pose(r, t)


def param_checksums(net):
    out = {}
    for k, v in net.state_dict().items():
        if k.startswith(("fusion_net", "exposure_embedding")):
            continue
        out[k] = np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])])
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    R, M, RU = import_reference()
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(1234)
    rnd = lambda *s: torch.rand(*s, generator=g)
    rndn = lambda *s: torch.randn(*s, generator=g)

    # ---- a1/a3 ray generation -------------------------------------------------
    poses = [pose((0.10, -0.20, 0.05), (0.10, 0.20, 0.30)), pose((1.2, 0.7, -2.1), (-1.5, 0.4, 2.0)),
             torch.cat([torch.eye(3), torch.zeros(3, 1)], 1)]
    d = {}
    for k, (H, W, f) in enumerate([(6, 8, 7.3), (5, 3, 262.75), (4, 4, 2.0)]):
        c2w = poses[k]
        o, dd = RU.get_rays(H, W, f, c2w)
        v = dd / torch.norm(dd, dim=-1, keepdim=True)
        d.update({f"hwf{k}": np.array([H, W, f]), f"c2w{k}": npy(c2w), f"rays_o{k}": npy(o), f"rays_d{k}": npy(dd),
                  f"viewdirs{k}": npy(v)})
    # a2 ndc
    o, dd = RU.get_rays(6, 8, 7.3, poses[0])
    o2, d2 = RU.ndc_rays(6, 8, 7.3, 1., o, dd)
    d.update(ndc_o=npy(o2), ndc_d=npy(d2))
    np.savez_compressed(os.path.join(OUT, "raygen.npz"), **d)

    # ---- a6 embedding -----------------------------------------------------------
    x = torch.cat([(rnd(40, 3) - .5) * 1.0, (rnd(40, 3) - .5) * 8.0, (rnd(16, 3) - .5) * 40.0], 0)
    e_xyz, _, _ = M.get_embedder(10, 0, -1)
    e_dir, _, _ = M.get_embedder(4, 0, -1)
    np.savez_compressed(os.path.join(OUT, "embed.npz"), x=npy(x), e63=npy(e_xyz(x)), e27=npy(e_dir(x)))

    # ---- a8 MLP -----------------------------------------------------------------
    d = {}
    for Wd, C in [(128, 128), (256, 16)]:
        coarse, fine = build_nets(M, Wd, C)
        tag = f"w{Wd}c{C}"
        for k, v in param_checksums(fine).items():
            d[f"{tag}.fine.{k}"] = v
        for k, v in param_checksums(coarse).items():
            d[f"{tag}.coarse.{k}"] = v
        pts = (rnd(48, 3) - .5) * 6.0
        dirs = rndn(48, 3)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        emb = torch.cat([e_xyz(pts), e_dir(dirs)], 1).requires_grad_()
        raw_full = fine(emb, output_transient=True)
        g_raw = rndn(*raw_full.shape)
        (g_emb,) = torch.autograd.grad(raw_full, emb, g_raw)
        raw_static = coarse(emb.detach(), output_transient=False)
        sig = coarse(emb.detach()[:, :63], sigma_only=True)
        d.update({f"{tag}.pts": npy(pts), f"{tag}.dirs": npy(dirs), f"{tag}.raw_full": npy(raw_full),
                  f"{tag}.g_raw": npy(g_raw), f"{tag}.g_emb": npy(g_emb), f"{tag}.raw_static": npy(raw_static),
                  f"{tag}.sigma": npy(sig)})
    np.savez_compressed(os.path.join(OUT, "mlp.npz"), **d)

    # ---- a9 compositing, variants A,B,C,D ----------------------------------------
    d = {}
    n, S, C = 12, 40, 5
    z = torch.sort(rnd(n, S) * 4.0, -1)[0]
    z[3] = torch.linspace(0, 4, S)
    raw = rndn(n, S, 3 + C + 6)
    sig_col, tsig_col = 3 + C, 3 + C + 4
    raw[..., sig_col] = torch.nn.functional.softplus(raw[..., sig_col] * 3)
    raw[..., tsig_col] = torch.nn.functional.softplus(raw[..., tsig_col])
    raw[..., 3 + C + 1:3 + C + 4] = torch.sigmoid(raw[..., 3 + C + 1:3 + C + 4])
    raw[..., -1] = torch.nn.functional.softplus(raw[..., -1])
    raw[0, :, sig_col] = 0.                       # sigma = 0 row
    raw[0, :, tsig_col] = 0.
    raw[1, :, sig_col] = 2000.                    # alpha saturates to exactly 1 everywhere
    raw[2, 10:14, sig_col] = 5000.                # saturated interior block ("surface")
    raw[4, :, sig_col] *= 50.
    d.update(z=npy(z), raw=npy(raw))
    ups = dict(g_rgb=rndn(n, 3), g_feat=rndn(n, C), g_disp=rndn(n) * 0.1, g_acc=rndn(n), g_depth=rndn(n),
               g_beta=rndn(n), g_w=rndn(n, S))
    d.update({k: npy(v) for k, v in ups.items()})

    def run(tag, raw_in, **kw):
        r = raw_in.clone().requires_grad_()
        rgb, feat, disp, acc, w, depth, tsig, beta = M.raw2outputs_NeRFH_NFF(r, z, **kw)
        loss = (acc * ups["g_acc"]).sum() + (w * ups["g_w"]).sum()
        d[f"{tag}.acc"], d[f"{tag}.weights"] = npy(acc), npy(w)
        if rgb is not None:
            loss = loss + (rgb * ups["g_rgb"]).sum() + (feat * ups["g_feat"]).sum() + (disp * ups["g_disp"]).sum() \
                + (depth * ups["g_depth"]).sum()
            d[f"{tag}.rgb"], d[f"{tag}.feat"], d[f"{tag}.disp"], d[f"{tag}.depth"] = npy(rgb), npy(feat), npy(disp), npy(depth)
            if beta is not None and beta.requires_grad:
                loss = loss + (beta * ups["g_beta"]).sum()
            if beta is not None:
                d[f"{tag}.beta"] = npy(beta)
        (g,) = torch.autograd.grad(loss, r)
        d[f"{tag}.g_raw"] = npy(g)

    run("A", raw, output_transient=True, beta_min=0.1, test_time=True, typ="fine", transient_at_test=True)
    run("Atrain", raw, output_transient=True, beta_min=0.1, test_time=False, typ="fine", transient_at_test=False)
    run("B", raw, output_transient=True, beta_min=0.1, test_time=True, typ="fine", transient_at_test=False)
    torch.manual_seed(0)
    run("C", raw[..., :3 + C + 1], output_transient=False, test_time=False, typ="coarse")
    torch.manual_seed(0)
    run("D", raw[..., sig_col:sig_col + 1], output_transient=False, test_time=True, typ="coarse")
    np.savez_compressed(os.path.join(OUT, "composite.npz"), **d)

    # ---- a10/a11 sample_pdf + merge -------------------------------------------------
    d = {}
    n, Nc = 16, 64
    zc = torch.linspace(0., 4., Nc).expand(n, Nc).contiguous()
    zc_j = torch.sort(zc + (rnd(n, Nc) - .5) * 0.05, -1)[0]
    w = rnd(n, Nc) * 0.05
    w[1] = 0.                                                         # all-zero weights
    w[2] = 0.; w[2, 20] = 1.                                          # one-hot
    w[3] = torch.exp(-0.5 * ((torch.arange(Nc) - 40.) / 1.5) ** 2)    # sharp surface
    w[4] = 1e-7 * rnd(Nc)
    w[5, :] = 0.; w[5, 1] = 0.5; w[5, -2] = 0.5                       # mass at both ends
    cap = {}
    real_ss = torch.searchsorted

    def spy(cdf, u, **kw):
        out = real_ss(cdf, u, **kw)
        cap["cdf"], cap["u"], cap["inds"] = cdf.clone(), u.clone(), out.clone()
        return out

    torch.searchsorted = spy
    try:
        for tag, zz, det, Ni in [("det128", zc, True, 128), ("det64", zc, True, 64), ("rand128", zc_j, False, 128)]:
            mid = .5 * (zz[..., 1:] + zz[..., :-1])
            torch.manual_seed(77)
            s = R.sample_pdf(mid, w[..., 1:-1], Ni, det=det)
            merged = torch.sort(torch.cat([zz, s], -1), -1)[0]
            d.update({f"{tag}.z": npy(zz), f"{tag}.samples": npy(s), f"{tag}.cdf": npy(cap["cdf"]),
                      f"{tag}.u": npy(cap["u"]), f"{tag}.inds": npy(cap["inds"]), f"{tag}.merged": npy(merged)})
    finally:
        torch.searchsorted = real_ss
    d["w"] = npy(w)
    np.savez_compressed(os.path.join(OUT, "sample_pdf.npz"), **d)

    # ---- a5 coarse depths (perturb) ---------------------------------------------------
    d = {}
    near, far = torch.full((5, 1), 0.5), torch.full((5, 1), 6.0)
    t = torch.linspace(0., 1., steps=16)
    d["z_lin"] = npy((near * (1. - t) + far * t).expand(5, 16))
    d["z_disp"] = npy((1. / (1. / near * (1. - t) + 1. / far * t)).expand(5, 16))
    np.savez_compressed(os.path.join(OUT, "depths.npz"), **d)

    # ---- FusionNet (post-render CNN) -------------------------------------
    d = {}
    coarse, _ = build_nets(M, 128, 16)
    for k, v in coarse.fusion_net.state_dict().items():
        d["sd." + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    rgb_in, feat_in = rnd(48, 3), rndn(48, 16)
    r_rgb, r_feat, fused = coarse.run_fusion_net(rgb_in.clone(), feat_in.clone(), 6, 8, 1)      # modules stay in train() mode
    d.update(rgb=npy(rgb_in), feat=npy(feat_in), render_rgb=npy(r_rgb), render_feat=npy(r_feat), fused=npy(fused))
    np.savez_compressed(os.path.join(OUT, "fusion.npz"), **d)

    # ---- end to end ----------------------------------------------------------------------
    d = {}
    hist = torch.full((1, 10), 10.)
    cases = [("ref_default", 128, 128, 64, True, 1.0, (6, 8)), ("metric", 256, 16, 128, True, 1.0, (6, 8)),
             ("metric_B", 256, 16, 128, False, 1.0, (4, 6)), ("surface", 256, 16, 128, True, 40.0, (6, 8)),
             ("ref_default_B", 128, 128, 64, False, 1.0, (4, 6))]
    for tag, Wd, C, Ni, tat, sscale, (H, W) in cases:
        coarse, fine = build_nets(M, Wd, C, sscale)
        for net in (coarse, fine):
            for prm in net.parameters():
                prm.requires_grad_(False)
        kw = make_kwargs(M, coarse, fine, 64, Ni, tat)
        focal = 525.505 * W / 640.
        c2w = poses[0].clone().requires_grad_()
        rgb, disp, acc, ex = R.render(H, W, focal, chunk=32768, c2w=c2w, near=0., far=4., img_idx=hist, **kw)
        feat = ex["feat_map"]
        loss = (feat ** 2).mean() + (rgb ** 2).mean()
        (g1,) = torch.autograd.grad(loss, c2w, retain_graph=True)
        gr, gf = rndn(*rgb.shape), rndn(*feat.shape)
        (g2,) = torch.autograd.grad((rgb * gr).sum() + (feat * gf).sum(), c2w)
        d.update({f"{tag}.cfg": np.array([Wd, C, Ni, int(tat), sscale, H, W, focal]), f"{tag}.c2w": npy(c2w),
                  f"{tag}.rgb": npy(rgb), f"{tag}.disp": npy(disp), f"{tag}.acc": npy(acc), f"{tag}.feat": npy(feat),
                  f"{tag}.g_c2w_loss": npy(g1), f"{tag}.g_rgb": npy(gr), f"{tag}.g_feat": npy(gf),
                  f"{tag}.g_c2w_lin": npy(g2)})
        print(tag, "loss", float(loss), "|g|", float(g1.abs().max()))
    np.savez_compressed(os.path.join(OUT, "end_to_end.npz"), **d)

    # ---- train mode: test_time=False, trainable weights, loss.backward() to the NeRF parameters (run_nefes.py:42-108) ----
    d = {}
    KEEP = ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight", "xyz_encoding_8.0.weight", "static_sigma.0.weight",
            "static_rgb.0.weight", "dir_encoding.0.weight", "xyz_encoding_final.weight", "transient_encoding.0.weight",
            "transient_encoding.4.weight", "transient_rgb.0.weight", "transient_sigma.0.weight", "transient_beta.0.weight")
    for tag, Wd, C, Nc, Ni, (H, W) in [("stage1", 128, 128, 32, 0, (4, 6)), ("full", 128, 128, 32, 32, (4, 6))]:
        coarse, fine = build_nets(M, Wd, C)
        kw = make_kwargs(M, coarse, fine, Nc, Ni, True, test_time=False, perturb=0.)
        focal = 525.505 * W / 640.
        rays_o, rays_d = RU.get_rays(H, W, focal, poses[0][:3, :4])
        rgb, disp, acc, ex = R.render(H, W, focal, chunk=32768, rays=(rays_o, rays_d), near=0., far=4., img_idx=hist, **kw)
        t_rgb, t_feat = torch.rand(H * W, 3), rndn(H * W, C)
        loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean()
        if "rgb0" in ex:
            loss = loss + ((ex["rgb0"] - t_rgb) ** 2).mean()
        loss.backward()
        d.update({f"{tag}.cfg": np.array([Wd, C, Nc, Ni, H, W, focal]), f"{tag}.c2w": npy(poses[0]), f"{tag}.t_rgb": npy(t_rgb),
                  f"{tag}.t_feat": npy(t_feat), f"{tag}.loss": np.array(float(loss)), f"{tag}.rgb": npy(rgb),
                  f"{tag}.disp": npy(disp), f"{tag}.acc": npy(acc)})
        for k, v in ex.items():
            d[f"{tag}.ex.{k}"] = npy(v)
        for name, net in (("coarse", coarse), ("fine", fine)):
            for k, prm in net.named_parameters():
                if prm.grad is not None and (k in KEEP or k.endswith(".bias")) and not k.startswith(("fusion", "exposure")):
                    d[f"{tag}.grad.{name}.{k}"] = npy(prm.grad)
        print(tag, "loss", float(loss), "extras", sorted(ex.keys()))
    np.savez_compressed(os.path.join(OUT, "train.npz"), **d)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
