"""Pin the CPU oracle (oracle/ref_cpu.py) against vectors captured from the real
reference by tools/make_goldens.py.  The oracle repeats the reference's torch op
sequence, so on the same torch build the fp32 results are bit-identical; the
asserts allow a few ulp so a different CPU/BLAS on the GPU box cannot flip them.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, rtol=2e-6, atol=2e-7):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_raygen(golden):
    g = golden("raygen")
    for k in range(3):
        H, W, f = g[f"hwf{k}"]
        o, d = O.ray_bundle(int(H), int(W), float(f), T(g[f"c2w{k}"]))
        assert np.array_equal(o.numpy(), g[f"rays_o{k}"])
        assert np.array_equal(d.numpy(), g[f"rays_d{k}"])
        v = d / torch.norm(d, dim=-1, keepdim=True)
        assert np.array_equal(v.numpy(), g[f"viewdirs{k}"])
    o, d = O.ray_bundle(6, 8, 7.3, T(g["c2w0"]))
    o2, d2 = O.ndc_warp(6, 8, 7.3, 1., o, d)
    assert np.array_equal(o2.numpy(), g["ndc_o"]) and np.array_equal(d2.numpy(), g["ndc_d"])


def test_embed(golden):
    g = golden("embed")
    x = T(g["x"])
    assert np.array_equal(O.freq_encode(x, 10).numpy(), g["e63"])
    assert np.array_equal(O.freq_encode(x, 4).numpy(), g["e27"])


def test_depths(golden):
    g = golden("depths")
    near, far = torch.full((5, 1), 0.5), torch.full((5, 1), 6.0)
    assert np.array_equal(O.coarse_depths(near, far, 16, False).numpy(), g["z_lin"])
    assert np.array_equal(O.coarse_depths(near, far, 16, True).numpy(), g["z_disp"])


@pytest.mark.parametrize("Wd,C", [(128, 128), (256, 16)])
def test_param_recipe_and_mlp(golden, Wd, C):
    g = golden("mlp")
    tag = f"w{Wd}c{C}"
    for typ in ("coarse", "fine"):
        p = O.make_field_params(typ, Wd, C)
        for k, v in p.items():
            chk = g[f"{tag}.{typ}.{k}"]
            got = np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])])
            np.testing.assert_allclose(got, chk, rtol=0, atol=0, err_msg=f"{typ}.{k}")
    pf, pc = O.make_field_params("fine", Wd, C), O.make_field_params("coarse", Wd, C)
    emb = torch.cat([O.freq_encode(T(g[f"{tag}.pts"]), 10), O.freq_encode(T(g[f"{tag}.dirs"]), 4)], 1).requires_grad_()
    raw = O.field_forward(pf, emb, output_transient=True)
    close(raw, g[f"{tag}.raw_full"], rtol=1e-5, atol=1e-6)
    (ge,) = torch.autograd.grad(raw, emb, T(g[f"{tag}.g_raw"]))
    close(ge, g[f"{tag}.g_emb"], rtol=1e-4, atol=1e-5)
    close(O.field_forward(pc, emb.detach(), output_transient=False), g[f"{tag}.raw_static"], rtol=1e-5, atol=1e-6)
    close(O.field_forward(pc, emb.detach()[:, :63], sigma_only=True), g[f"{tag}.sigma"], rtol=1e-5, atol=1e-6)


VARIANTS = {
    "A": dict(output_transient=True, test_time=True, typ="fine", transient_at_test=True),
    "Atrain": dict(output_transient=True, test_time=False, typ="fine", transient_at_test=False),
    "B": dict(output_transient=True, test_time=True, typ="fine", transient_at_test=False),
    "C": dict(output_transient=False, test_time=False, typ="coarse"),
    "D": dict(output_transient=False, test_time=True, typ="coarse"),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_composite(golden, tag):
    g = golden("composite")
    C = g["g_feat"].shape[1]
    raw = T(g["raw"])
    if tag == "C":
        raw = raw[..., :3 + C + 1]
    if tag == "D":
        raw = raw[..., 3 + C:3 + C + 1]
    raw = raw.clone().requires_grad_()
    z = T(g["z"])
    out = O.composite(raw, z, **VARIANTS[tag])
    close(out.acc, g[f"{tag}.acc"], rtol=1e-6)
    close(out.weights, g[f"{tag}.weights"], rtol=1e-6)
    loss = (out.acc * T(g["g_acc"])).sum() + (out.weights * T(g["g_w"])).sum()
    if out.rgb is not None:
        close(out.rgb, g[f"{tag}.rgb"], rtol=1e-6, atol=1e-6)
        close(out.feat, g[f"{tag}.feat"], rtol=1e-6, atol=1e-6)
        close(out.disp, g[f"{tag}.disp"], rtol=1e-6)
        close(out.depth, g[f"{tag}.depth"], rtol=1e-6, atol=1e-6)
        close(out.beta, g[f"{tag}.beta"], rtol=1e-6, atol=1e-6)
        loss = loss + (out.rgb * T(g["g_rgb"])).sum() + (out.feat * T(g["g_feat"])).sum() \
            + (out.disp * T(g["g_disp"])).sum() + (out.depth * T(g["g_depth"])).sum()
        if out.beta.requires_grad:
            loss = loss + (out.beta * T(g["g_beta"])).sum()
    (gr,) = torch.autograd.grad(loss, raw)
    # row 0 has sigma == 0 everywhere: the reference itself yields disp = 0/0 = NaN there (and NaN
    # sigma-gradients through it); every other row, including the alpha == 1 saturated ones, is finite.
    assert np.isfinite(gr.numpy()[1:]).all()
    close(gr, g[f"{tag}.g_raw"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,det,Ni", [("det128", True, 128), ("det64", True, 64), ("rand128", False, 128)])
def test_sample_pdf(golden, tag, det, Ni):
    g = golden("sample_pdf")
    z, w = T(g[f"{tag}.z"]), T(g["w"])
    mid = .5 * (z[..., 1:] + z[..., :-1])
    u = None if det else T(g[f"{tag}.u"])
    s, cdf, inds = O.inverse_cdf_samples(mid, w[..., 1:-1], Ni, det=det, u=u, return_debug=True)
    assert np.array_equal(cdf.numpy(), g[f"{tag}.cdf"])          # same torch ops => same bits
    assert np.array_equal(inds.numpy(), g[f"{tag}.inds"])        # indices bit-exact
    assert np.array_equal(s.numpy(), g[f"{tag}.samples"])
    merged = torch.sort(torch.cat([z, s], -1), -1)[0]
    assert np.array_equal(merged.numpy(), g[f"{tag}.merged"])


@pytest.mark.parametrize("tag", ["ref_default", "metric", "metric_B", "surface", "ref_default_B"])
def test_end_to_end(golden, tag):
    g = golden("end_to_end")
    Wd, C, Ni, tat, sscale, H, W, focal = g[f"{tag}.cfg"]
    Wd, C, Ni, H, W = int(Wd), int(C), int(Ni), int(H), int(W)
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    for p in (pc, pf):
        p["static_sigma.0.weight"] = p["static_sigma.0.weight"] * float(sscale)
        p["static_sigma.0.bias"] = p["static_sigma.0.bias"] * float(sscale)
    cfg = O.RenderCfg(N_samples=64, N_importance=Ni, transient_at_test=bool(tat))
    c2w = T(g[f"{tag}.c2w"]).clone().requires_grad_()
    rgb, disp, acc, ex = O.render(H, W, float(focal), pc, pf, cfg, c2w=c2w, near=0., far=4.,
                                  hist=torch.full((1, 10), 10.))
    feat = ex["feat_map"]
    close(rgb, g[f"{tag}.rgb"], rtol=1e-5, atol=1e-6)
    close(feat, g[f"{tag}.feat"], rtol=1e-5, atol=1e-6)
    close(disp, g[f"{tag}.disp"], rtol=1e-5)
    close(acc, g[f"{tag}.acc"], rtol=1e-5)
    (g1,) = torch.autograd.grad(O.bench_loss(rgb, feat), c2w, retain_graph=True)
    (g2,) = torch.autograd.grad((rgb * T(g[f"{tag}.g_rgb"])).sum() + (feat * T(g[f"{tag}.g_feat"])).sum(), c2w)
    # same op graph as the reference: agreement is far inside the fp32 noise floor of this gradient
    sc1, sc2 = np.abs(g[f"{tag}.g_c2w_loss"]).max(), np.abs(g[f"{tag}.g_c2w_lin"]).max()
    assert np.abs(g1.numpy() - g[f"{tag}.g_c2w_loss"]).max() <= 2e-5 * sc1
    assert np.abs(g2.numpy() - g[f"{tag}.g_c2w_lin"]).max() <= 2e-5 * sc2


@pytest.mark.parametrize("tag", ["stage1", "full"])
def test_train_mode(golden, tag):
    """Train mode of the reference (test_time=False, perturb=0, trainable weights): maps, train extras
    (rendering.py:160-173) and loss.backward() to the NeRF parameters (run_nefes.py:42-108)."""
    g = golden("train")
    Wd, C, Nc, Ni, H, W, focal = g[f"{tag}.cfg"]
    Wd, C, Nc, Ni, H, W = int(Wd), int(C), int(Nc), int(Ni), int(H), int(W)
    pc = {k: v.requires_grad_() for k, v in O.make_field_params("coarse", Wd, C).items()}
    pf = {k: v.requires_grad_() for k, v in O.make_field_params("fine", Wd, C).items()}
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, transient_at_test=True)
    rays_o, rays_d = O.ray_bundle(H, W, float(focal), T(g[f"{tag}.c2w"])[:3, :4])
    rgb, disp, acc, ex = O.render(H, W, float(focal), pc, pf, cfg, rays=(rays_o, rays_d), near=0., far=4.,
                                  hist=torch.full((1, 10), 10.))
    close(rgb, g[f"{tag}.rgb"], rtol=1e-5, atol=1e-6)
    close(acc, g[f"{tag}.acc"], rtol=1e-5)
    for k in [k for k in g if k.startswith(f"{tag}.ex.")]:
        close(ex[k.split(".ex.")[1]], g[k], rtol=2e-5, atol=2e-6)
    t_rgb, t_feat = T(g[f"{tag}.t_rgb"]), T(g[f"{tag}.t_feat"])
    loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean()
    if Ni > 0:
        loss = loss + ((ex["rgb0"] - t_rgb) ** 2).mean()
    assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-6 * float(g[f"{tag}.loss"])
    loss.backward()
    n = 0
    for k in [k for k in g if k.startswith(f"{tag}.grad.")]:
        _, _, net, name = k.split(".", 3)
        got = (pc if net == "coarse" else pf)[name].grad
        assert got is not None, k
        assert np.abs(got.numpy() - g[k]).max() <= 2e-5 * max(np.abs(g[k]).max(), 1e-12), k
        n += 1
    assert n >= 19


# ---- round-4 fixtures (tools/make_golden_shapes.py -> shapes.npz): W = 256 with the reference's FEATURE_DIM = 128, W = 128 with 16
# ---- feature channels, and more than 256 samples per ray ---------------------------------------------------------------------
@pytest.mark.parametrize("Wd,C", [(256, 128), (128, 16)])
def test_shapes_param_recipe_and_mlp(golden, Wd, C):
    g = golden("shapes")
    tag = f"mlp.w{Wd}c{C}"
    for typ in ("coarse", "fine"):
        for k, v in O.make_field_params(typ, Wd, C).items():
            got = np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])])
            np.testing.assert_allclose(got, g[f"{tag}.{typ}.{k}"], rtol=0, atol=0, err_msg=f"{typ}.{k}")
    pf, pc = O.make_field_params("fine", Wd, C), O.make_field_params("coarse", Wd, C)
    emb = torch.cat([O.freq_encode(T(g[f"{tag}.pts"]), 10), O.freq_encode(T(g[f"{tag}.dirs"]), 4)], 1).requires_grad_()
    raw = O.field_forward(pf, emb, output_transient=True)
    close(raw, g[f"{tag}.raw_full"], rtol=1e-5, atol=1e-6)
    (ge,) = torch.autograd.grad(raw, emb, T(g[f"{tag}.g_raw"]))
    close(ge, g[f"{tag}.g_emb"], rtol=1e-4, atol=1e-5)
    close(O.field_forward(pc, emb.detach(), output_transient=False), g[f"{tag}.raw_static"], rtol=1e-5, atol=1e-6)
    close(O.field_forward(pc, emb.detach()[:, :63], sigma_only=True), g[f"{tag}.sigma"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["w256c128", "w128c16", "w256c128_B", "s320", "s384"])
def test_shapes_end_to_end(golden, tag):
    g = golden("shapes")
    t = f"e2e.{tag}"
    Wd, C, Nc, Ni, tat, H, W, focal = g[f"{t}.cfg"]
    Wd, C, Nc, Ni, H, W = int(Wd), int(C), int(Nc), int(Ni), int(H), int(W)
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, transient_at_test=bool(tat))
    c2w = T(g[f"{t}.c2w"]).clone().requires_grad_()
    rgb, disp, acc, ex = O.render(H, W, float(focal), pc, pf, cfg, c2w=c2w, near=0., far=4., hist=torch.full((1, 10), 10.))
    feat = ex["feat_map"]
    close(rgb, g[f"{t}.rgb"], rtol=1e-5, atol=1e-6)
    close(feat, g[f"{t}.feat"], rtol=1e-5, atol=1e-6)
    close(disp, g[f"{t}.disp"], rtol=1e-5)
    close(acc, g[f"{t}.acc"], rtol=1e-5)
    (g1,) = torch.autograd.grad(O.bench_loss(rgb, feat), c2w, retain_graph=True)
    (g2,) = torch.autograd.grad((rgb * T(g[f"{t}.g_rgb"])).sum() + (feat * T(g[f"{t}.g_feat"])).sum(), c2w)
    assert np.abs(g1.numpy() - g[f"{t}.g_c2w_loss"]).max() <= 2e-5 * np.abs(g[f"{t}.g_c2w_loss"]).max()
    assert np.abs(g2.numpy() - g[f"{t}.g_c2w_lin"]).max() <= 2e-5 * np.abs(g[f"{t}.g_c2w_lin"]).max()


@pytest.mark.parametrize("tag", ["w256c128", "w128c16"])
def test_shapes_train_mode(golden, tag):
    g = golden("shapes")
    t = f"train.{tag}"
    Wd, C, Nc, Ni, H, W, focal = g[f"{t}.cfg"]
    Wd, C, Nc, Ni, H, W = int(Wd), int(C), int(Nc), int(Ni), int(H), int(W)
    pc = {k: v.requires_grad_() for k, v in O.make_field_params("coarse", Wd, C).items()}
    pf = {k: v.requires_grad_() for k, v in O.make_field_params("fine", Wd, C).items()}
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, transient_at_test=True)
    rays_o, rays_d = O.ray_bundle(H, W, float(focal), T(g[f"{t}.c2w"])[:3, :4])
    rgb, disp, acc, ex = O.render(H, W, float(focal), pc, pf, cfg, rays=(rays_o, rays_d), near=0., far=4., hist=torch.full((1, 10), 10.))
    close(rgb, g[f"{t}.rgb"], rtol=1e-5, atol=1e-6)
    for k in [k for k in g if k.startswith(f"{t}.ex.")]:
        close(ex[k.split(".ex.")[1]], g[k], rtol=2e-5, atol=2e-6)
    t_rgb, t_feat = T(g[f"{t}.t_rgb"]), T(g[f"{t}.t_feat"])
    loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean() + ((ex["rgb0"] - t_rgb) ** 2).mean()
    assert abs(float(loss.detach()) - float(g[f"{t}.loss"])) < 1e-6 * float(g[f"{t}.loss"])
    loss.backward()
    n = 0
    for k in [k for k in g if k.startswith(f"{t}.grad.")]:
        _, _, _, net, name = k.split(".", 4)
        got = (pc if net == "coarse" else pf)[name].grad
        assert got is not None, k
        assert np.abs(got.numpy() - g[k]).max() <= 2e-5 * max(np.abs(g[k]).max(), 1e-12), k
        n += 1
    assert n >= 19


# ---- embeddings with fewer octaves (tools/make_golden_reduced.py -> reduced.npz): --reduce_embedding 0 / 1 and smaller --multires
# ---- (nerfh_nff.py:303-354); the networks are sized by the embedder's out_dim (:633-659) ------------------------------------------
def _reduced_cfg(g, t):
    Wd, C, mr, mrv, mode, Nc, Ni, H, W, focal, in_xyz, in_dir = g[f"{t}.cfg"]
    n_xyz, n_dir = (int(in_xyz) - 3) // 6, (int(in_dir) - 3) // 6
    # the reference's rule for the octave counts (get_embedder): mode 0 halves, mode 1 drops them, otherwise multires itself
    assert n_xyz == {0: int(mr) // 2, 1: 0}.get(int(mode), int(mr)) and n_dir == {0: int(mrv) // 2, 1: 0}.get(int(mode), int(mrv))
    return int(Wd), int(C), int(Nc), int(Ni), int(H), int(W), float(focal), int(in_xyz), int(in_dir), n_xyz, n_dir


@pytest.mark.parametrize("tag", ["mode0", "mode1", "m6v2", "mode0_w256"])
def test_reduced_embedding_end_to_end(golden, tag):
    g = golden("reduced")
    t = f"e2e.{tag}"
    Wd, C, Nc, Ni, H, W, focal, in_xyz, in_dir, n_xyz, n_dir = _reduced_cfg(g, t)
    pc, pf = O.make_field_params("coarse", Wd, C, in_xyz=in_xyz, in_dir=in_dir), O.make_field_params("fine", Wd, C, in_xyz=in_xyz, in_dir=in_dir)
    for typ, p in (("coarse", pc), ("fine", pf)):
        for k, v in p.items():
            got = np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])])
            np.testing.assert_allclose(got, g[f"{t}.{typ}.{k}"], rtol=0, atol=0, err_msg=f"{typ}.{k}")
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, n_freq_xyz=n_xyz, n_freq_dir=n_dir)
    c2w = T(g[f"{t}.c2w"]).clone().requires_grad_()
    rgb, disp, acc, ex = O.render(H, W, focal, pc, pf, cfg, c2w=c2w, near=0., far=4., hist=torch.full((1, 10), 10.))
    feat = ex["feat_map"]
    close(rgb, g[f"{t}.rgb"], rtol=1e-5, atol=1e-6)
    close(feat, g[f"{t}.feat"], rtol=1e-5, atol=1e-6)
    close(disp, g[f"{t}.disp"], rtol=1e-5)
    close(acc, g[f"{t}.acc"], rtol=1e-5)
    (g1,) = torch.autograd.grad(O.bench_loss(rgb, feat), c2w, retain_graph=True)
    (g2,) = torch.autograd.grad((rgb * T(g[f"{t}.g_rgb"])).sum() + (feat * T(g[f"{t}.g_feat"])).sum(), c2w)
    assert np.abs(g1.numpy() - g[f"{t}.g_c2w_loss"]).max() <= 2e-5 * np.abs(g[f"{t}.g_c2w_loss"]).max()
    assert np.abs(g2.numpy() - g[f"{t}.g_c2w_lin"]).max() <= 2e-5 * np.abs(g[f"{t}.g_c2w_lin"]).max()


def test_reduced_embedding_train_mode(golden):
    g = golden("reduced")
    t = "train.mode0"
    Wd, C, Nc, Ni, H, W, focal, in_xyz, in_dir = g[f"{t}.cfg"]
    Wd, C, Nc, Ni, H, W, in_xyz, in_dir = (int(v) for v in (Wd, C, Nc, Ni, H, W, in_xyz, in_dir))
    pc = {k: v.requires_grad_() for k, v in O.make_field_params("coarse", Wd, C, in_xyz=in_xyz, in_dir=in_dir).items()}
    pf = {k: v.requires_grad_() for k, v in O.make_field_params("fine", Wd, C, in_xyz=in_xyz, in_dir=in_dir).items()}
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, n_freq_xyz=(in_xyz - 3) // 6, n_freq_dir=(in_dir - 3) // 6)
    rays_o, rays_d = O.ray_bundle(H, W, float(focal), T(g[f"{t}.c2w"])[:3, :4])
    rgb, disp, acc, ex = O.render(H, W, float(focal), pc, pf, cfg, rays=(rays_o, rays_d), near=0., far=4., hist=torch.full((1, 10), 10.))
    close(rgb, g[f"{t}.rgb"], rtol=1e-5, atol=1e-6)
    close(ex["feat_map"], g[f"{t}.feat"], rtol=1e-5, atol=1e-6)
    t_rgb, t_feat = T(g[f"{t}.t_rgb"]), T(g[f"{t}.t_feat"])
    loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean() + ((ex["rgb0"] - t_rgb) ** 2).mean()
    assert abs(float(loss.detach()) - float(g[f"{t}.loss"])) < 1e-6 * float(g[f"{t}.loss"])
    loss.backward()
    n = 0
    for k in [k for k in g if k.startswith(f"{t}.grad.")]:
        _, _, _, net, name = k.split(".", 4)
        got = (pc if net == "coarse" else pf)[name].grad
        assert got is not None and tuple(got.shape) == g[k].shape, k
        assert np.abs(got.numpy() - g[k]).max() <= 2e-5 * max(np.abs(g[k]).max(), 1e-12), k
        n += 1
    assert n >= 19
