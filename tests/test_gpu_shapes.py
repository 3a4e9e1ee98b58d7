"""Network shapes and sample counts beyond the two canonical instances (VERDICT r3 "What's missing" 1, 2).

The reference's FEATURE_DIM is a module constant, 128 (script/models/nerfh_nff.py:21, default f_dim :427, never overridden by
create_nerf :645-659), while --netwidth is a free flag (script/models/options.py:30-31): `NeRFH_NFF(W=256)` therefore has C = 128.
The fp16 two-part field kernels are compiled per (width, head class) -- csrc/layout.h nefes_head_class -- and take C at run time;
fixtures: tests/golden/shapes.npz, captured from the reference by tools/make_golden_shapes.py.  N_samples + N_importance beyond 256
(rendering.py:132-141 takes any): the compositor's one-sample-per-lane kernels in up to eight passes.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import branch as B
from tests import parity_log as P
from tests.test_gpu_parity import DEV, T, _modules, check_end_to_end, rel
from tests.test_gpu_train import check_field_train_weight_grads, check_train_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Wd,C", [(256, 128), (128, 16)])
def test_module_init_and_mlp_vs_reference(golden, Wd, C):
    """shapes.npz `mlp.*`: parameter checksums, then raw outputs and the gradient to the embedded input at 48 points, one call per
    point (the fixture used one direction per point; run_network expands one direction per ray)."""
    from nefes_amd.field import run_network_NeRFH_NFF
    g = golden("shapes")
    tag = f"mlp.w{Wd}c{C}"
    coarse, fine = _modules(Wd, C)
    for typ, m in (("coarse", coarse), ("fine", fine)):
        for k, v in m.state_dict().items():
            key = f"{tag}.{typ}.{k}"
            if key in g:
                v = v.cpu()
                np.testing.assert_allclose(np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.flatten()[0])]),
                                           g[key], rtol=0, atol=0, err_msg=key)
    pts, dirs = T(g[f"{tag}.pts"]).to(DEV), T(g[f"{tag}.dirs"])
    worst = 0.
    for idx in range(0, 48, 5):
        p1 = pts[idx:idx + 1].reshape(1, 1, 3).clone().requires_grad_()
        v1 = dirs[idx:idx + 1].to(DEV).clone().requires_grad_()
        raw = run_network_NeRFH_NFF(p1, v1, None, fine, typ='fine', output_transient=True, test_time=True)
        np.testing.assert_allclose(raw.detach().cpu().numpy()[0, 0], g[f"{tag}.raw_full"][idx], rtol=1e-4, atol=2e-6)
        worst = max(worst, rel(raw[0, 0], g[f"{tag}.raw_full"][idx]))
        raw.backward(T(g[f"{tag}.g_raw"][idx]).to(DEV).reshape(1, 1, -1))
        pe = T(g[f"{tag}.pts"][idx:idx + 1]).double().requires_grad_()
        de = dirs[idx:idx + 1].double().requires_grad_()
        emb = torch.cat([O.freq_encode(pe, 10), O.freq_encode(de, 4)], 1)
        emb.backward(T(g[f"{tag}.g_emb"][idx:idx + 1]).double())
        assert rel(p1.grad.reshape(1, 3), pe.grad) < 2e-4, (idx, p1.grad, pe.grad)
        assert rel(v1.grad, de.grad) < 2e-4
    P.record(f"shapes_mlp[{Wd},{C}]", "raw outputs vs reference fixture (worst of 10 points)", e_hip=worst, e_ref=None, bound=1e-4)
    N, S = 2, 24
    sig = run_network_NeRFH_NFF(pts.reshape(N, S, 3), dirs[:N].to(DEV), None, coarse, typ='coarse', output_transient=False, test_time=True)
    np.testing.assert_allclose(sig.cpu().numpy().reshape(-1), g[f"{tag}.sigma"].reshape(-1), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("tag", ["w256c128", "w128c16", "w256c128_B", "s320", "s384"])
def test_render_end_to_end_vs_reference(golden, tag):
    g = golden("shapes")
    Wd, C, Nc, Ni, tat, H, W, focal = g[f"e2e.{tag}.cfg"]
    check_end_to_end(g, f"e2e.{tag}", int(Wd), int(C), int(Nc), int(Ni), bool(tat), 1.0, int(H), int(W), float(focal))


@pytest.mark.parametrize("Wd,C,S", [(256, 128, 192), (128, 16, 128), (256, 64, 72), (128, 141, 40), (256, 5, 64), (128, 30, 33)])
def test_field_from_rays_vs_oracle(Wd, C, S):
    """C as a run-time parameter inside the head classes (class 0: 3 + C <= 32, class 1: <= 144), at both widths, ragged tile counts:
    raw outputs three-way (HIP, fp32 oracle, float64), gradients to the rays on the kernels' own ReLU branch pattern."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    coarse, fine = _modules(Wd, C)
    pf = O.make_field_params("fine", Wd, C)
    gen = torch.Generator().manual_seed(23)
    N = 7
    o = (torch.rand(N, 3, generator=gen) - .5)
    d = torch.randn(N, 3, generator=gen)
    v = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(N, S, generator=gen) * 4, -1)[0]
    g_raw = torch.randn(N, S, 3 + C + 6, generator=gen)
    res = {}
    for dt in (torch.float64, torch.float32):
        oo, dd, vv = (t.to(dt).clone().requires_grad_() for t in (o, d, v))
        pts = oo[:, None, :] + dd[:, None, :] * z.to(dt)[..., None]
        raw = O.query_field({k: w.to(dt) for k, w in pf.items()}, pts, vv, "fine", True, True)
        raw.backward(g_raw.to(dt))
        res[dt] = (raw.detach(), oo.grad, dd.grad, vv.grad)
    oh, dh, vh = (t.to(DEV).clone().requires_grad_() for t in (o, d, v))
    with B.tapped() as tap:
        raw_t = ops.FieldFromRays.apply(oh, dh, vh, z.to(DEV), fine.packed(), L.FIELD_FULL)
    raw_t.backward(g_raw.permute(0, 2, 1).contiguous().to(DEV))
    tag = f"shapes_field_from_rays[{Wd},{C},{S}]"
    B.three_way(tag, "raw", raw_t.permute(0, 2, 1), res[torch.float32][0], res[torch.float64][0])

    def oracle_run(dt, act, _):
        oo, dd, vv = (t.to(dt).clone().requires_grad_() for t in (o, d, v))
        pts = oo[:, None, :] + dd[:, None, :] * z.to(dt)[..., None]
        O.query_field({k: w.to(dt) for k, w in pf.items()}, pts, vv, "fine", True, True, act=act).backward(g_raw.to(dt))
        return {"d rays_o": oo.grad, "d rays_d": dd.grad, "d viewdirs": vv.grad}

    B.pinned_gradients(tag, {"d rays_o": oh.grad, "d rays_d": dh.grad, "d viewdirs": vh.grad}, tap, Wd, oracle_run)
    # the sigma-only pass of the coarse network (one instance per width, whatever C)
    sig = ops.FieldFromRays.apply(o.to(DEV), d.to(DEV), v.to(DEV), z.to(DEV), coarse.packed(), L.FIELD_SIGMA)
    pc = O.make_field_params("coarse", Wd, C)
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    ref = O.query_field({k: w.double() for k, w in pc.items()}, pts.double(), v.double(), "coarse", False, True)
    assert rel(sig[:, 0, :], ref[..., 0]) < 2e-5


@pytest.mark.parametrize("Wd,C,typ", [(256, 128, "fine"), (256, 128, "coarse"), (128, 16, "fine"), (128, 16, "coarse"), (256, 64, "fine")])
def test_field_train_weight_grads(Wd, C, typ, monkeypatch):
    """Train mode at the new shapes: saved pre-activations and every parameter gradient against the float64 oracle on the kernels'
    branch pattern (the fp16 two-part TRAIN instances; the fp32-MFMA pipe exists for the canonical shapes only)."""
    check_field_train_weight_grads(Wd, C, typ, "h3", monkeypatch)


@pytest.mark.parametrize("tag", ["w256c128", "w128c16"])
def test_train_mode_vs_reference_golden(golden, tag):
    check_train_golden(golden("shapes"), f"train.{tag}")


def test_device_repack_bit_identical_at_the_new_shapes():
    from tests.test_gpu_train import test_device_repack_bit_identical as chk
    chk(256, 128, "fine", 63)
    chk(128, 16, "coarse", 63)


def test_unserved_shapes_fail_loudly_naming_the_compiled_set(monkeypatch):
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    with pytest.raises(RuntimeError, match="Compiled: fp16 two-part instances"):
        NeRFH_NFF('fine', W=256, f_dim=150, encode_appearance=True, encode_transient=True).to(DEV).packed()
    with pytest.raises(RuntimeError, match="Compiled: fp16 two-part instances"):
        NeRFH_NFF('fine', W=192, f_dim=16, encode_appearance=True, encode_transient=True).to(DEV).packed()
    # nefes_blob_info itself refuses what no kernel serves (it used to accept (256, 128) and fail at the launch)
    info = L.NefesBlobInfo()
    assert L.load().nefes_blob_info(L.NefesNetDesc(256, 142, 1, 0), info) == -2
    assert L.load().nefes_blob_info(L.NefesNetDesc(128, 16, 1, 1), info) == 0          # packs (fp32 streams) ...
    _, fine = _modules(256, 128)
    o = torch.zeros(4, 3, device=DEV)
    d = torch.nn.functional.normalize(torch.ones(4, 3, device=DEV), dim=-1)
    z = torch.linspace(0.5, 3., 8, device=DEV).expand(4, 8).contiguous()
    monkeypatch.setattr(ops, "SPLIT", "x6")                                             # ... but only the fp16 instances exist for it
    with pytest.raises(RuntimeError, match="Compiled: fp16 two-part instances"):
        ops.FieldFromRays.apply(o, d, d, z, fine.packed(), L.FIELD_FULL)


def test_composite_above_256_samples_matches_oracle():
    """S = 320, 384, 500, 512 (one sample per lane, five to eight passes) x variants A and D against the float64 oracle."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    gen = torch.Generator().manual_seed(31)
    for S in (320, 384, 500, 512):
        N, C = 9, 16
        z = torch.sort(torch.rand(N, S, generator=gen) * 4, -1)[0]
        raw = torch.randn(N, S, 3 + C + 6, generator=gen)
        raw[..., 3 + C] = torch.nn.functional.softplus(raw[..., 3 + C] * 3) * 0.3
        raw[..., 3 + C + 4] = torch.nn.functional.softplus(raw[..., 3 + C + 4]) * 0.3
        raw[..., 3 + C + 1:3 + C + 4] = torch.sigmoid(raw[..., 3 + C + 1:3 + C + 4])
        raw[..., -1] = torch.nn.functional.softplus(raw[..., -1])
        raw[1, :, 3 + C] = 0.
        raw[2, 100:104, 3 + C] = 5000.
        g_rgb, g_feat = torch.randn(N, 3, generator=gen), torch.randn(N, C, generator=gen)
        r64 = raw.double().requires_grad_()
        ref = O.composite(r64, z.double(), output_transient=True, test_time=True, typ="fine", transient_at_test=True)
        ((ref.rgb * g_rgb.double()).sum() + (ref.feat * g_feat.double()).sum()).backward()
        rt = raw.permute(0, 2, 1).contiguous().to(DEV).requires_grad_()
        rgb, feat, disp, acc, depth, w, beta = ops.Composite.apply(rt, z.to(DEV), C, L.COMP_TRANSIENT, 0.1)
        ((rgb * g_rgb.to(DEV)).sum() + (feat * g_feat.to(DEV)).sum()).backward()
        e = {"rgb": rel(rgb, ref.rgb), "feat": rel(feat, ref.feat), "acc": rel(acc, ref.acc), "weights": rel(w, ref.weights),
             "d raw": rel(rt.grad.permute(0, 2, 1), r64.grad)}
        P.record(f"composite_S[{S}]", "variant A maps and d raw vs float64", e_hip=max(e.values()), e_ref=None, bound=2e-5)
        assert max(e.values()) < 2e-5, (S, e)
        _, _, _, acc_d, _, w_d, _ = ops.composite_fwd(rt.detach()[:, 3 + C:3 + C + 1, :].contiguous(), z.to(DEV), C, L.COMP_SIGMA_ONLY)
        ref_d = O.composite(raw[..., 3 + C:3 + C + 1].double(), z.double(), output_transient=False, test_time=True, typ="coarse")
        assert rel(w_d, ref_d.weights) < 2e-6 and rel(acc_d, ref_d.acc) < 2e-6


def test_full_frame_640x480_properties_at_w256_c128():
    """The BASELINE frame (640x480, 64+128 samples) with the network the reference builds at --netwidth 256 (C = 128: 59 M fine samples,
    ~90 GB resident): the size-independent properties of tests/test_gpu_parity.py::test_full_frame_640x480_properties -- finite maps,
    compositing weights a sub-probability, bit-identical rerun, row shards equal the rows of the full frame bit for bit, pose gradient
    linear in the loss and additive over shards."""
    from nefes_amd import dist as D
    from tests.test_gpu_parity import _dropin, _kwargs
    R, M = _dropin()
    coarse, fine = _modules(256, 128)
    kw = _kwargs(M, coarse, fine, 128, True)
    H, W, f = 480, 640, 525.505
    c2w = O.bench_pose().to(DEV).requires_grad_()
    rgb, disp, acc, ex = R.render(H, W, f, c2w=c2w, near=0., far=4., **kw)
    feat = ex["feat_map"]
    assert rgb.shape == (H * W, 3) and feat.shape == (H * W, 128)
    assert torch.isfinite(rgb).all() and torch.isfinite(feat).all() and torch.isfinite(disp).all()
    assert (acc > 0).all() and (acc <= 1 + 1e-5).all()
    loss = O.bench_loss(rgb, feat)
    (g1,) = torch.autograd.grad(loss, c2w, retain_graph=True)
    (g3,) = torch.autograd.grad(3.0 * loss, c2w)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0 and rel(g3, 3.0 * g1) < 1e-6
    chk = (float(rgb.double().sum()), float(feat.double().sum()))
    del rgb, disp, acc, ex, loss
    torch.cuda.empty_cache()
    rgb2, _, _, ex2 = R.render(H, W, f, c2w=c2w, near=0., far=4., **kw)
    assert (float(rgb2.double().sum()), float(ex2["feat_map"].double().sum())) == chk
    acc_g = torch.zeros_like(g1)
    for rank in range(2):
        row0, n = D.row_shard(H, rank, 2)
        r, _, _, e = R.render(H, W, f, c2w=c2w, near=0., far=4., row_range=(row0, n), **kw)
        assert torch.equal(r, rgb2[row0 * W:(row0 + n) * W]) and torch.equal(e["feat_map"], ex2["feat_map"][row0 * W:(row0 + n) * W])
        part = (e["feat_map"] ** 2).sum() / (H * W * 128) + (r ** 2).sum() / (H * W * 3)
        acc_g += torch.autograd.grad(part, c2w)[0]
        del r, e, part
    assert rel(acc_g, g1) < 1e-5
