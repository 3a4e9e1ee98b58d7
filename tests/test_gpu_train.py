"""Train mode: forward with saved pre-activations + weight-gradient kernels vs the oracle's autograd.

Ground truth = oracle/ref_cpu.py evaluated in float64 (the reference's fp32 autograd is itself ~1e-5 away from it).
Tolerance: every parameter gradient within 2e-4 of its own max-norm (fp32 products accumulated over up to ~1e5 samples).
"""
import types

import pytest
import torch

from oracle import ref_cpu as O
from tests import branch as B
from tests import parity_log as P

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(typ, Wd, C):
    from nefes_amd.field import NeRFH_NFF
    if typ == "coarse":
        return NeRFH_NFF('coarse', W=Wd, f_dim=C).to(DEV)
    return NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).to(DEV)


def _oracle_params(net, names, dtype=torch.float64):
    sd = dict(net.named_parameters())
    p = {}
    for n, t in sd.items():
        if n.startswith(("fusion_net", "exposure_embedding")):
            continue
        p[n] = t.detach().cpu().to(dtype).clone().requires_grad_(n in names)
    return p


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("pipe", ["h3", "f32"])
@pytest.mark.parametrize("Wd,C", [(256, 16), (128, 128)])
@pytest.mark.parametrize("typ", ["coarse", "fine"])
def test_field_train_weight_grads(Wd, C, typ, pipe, monkeypatch):
    """pipe: the train-mode forward and the fused dX chain on the fp16 two-part instances (default) or on the fp32-MFMA ones."""
    check_field_train_weight_grads(Wd, C, typ, pipe, monkeypatch)


def check_field_train_weight_grads(Wd, C, typ, pipe, monkeypatch):
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd import train as TR
    monkeypatch.setattr(ops, "SPLIT", pipe)
    monkeypatch.setattr(ops, "TIMERS", {})
    torch.manual_seed(11)
    N, S = 37, 24                                                   # 888 samples: 7 tiles, the last one ragged
    mode = L.FIELD_STATIC if typ == "coarse" else L.FIELD_FULL
    net = _net(typ, Wd, C)
    g = torch.Generator().manual_seed(2)
    rays_o = torch.randn(N, 3, generator=g) * 0.3
    rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * 3.5 + 0.2, -1)[0]
    TR.DEBUG = {}
    try:
        with B.tapped() as tap:
            raw_t = TR.field_train(net, mode, rays_o.to(DEV), rays_d.to(DEV), rays_d.to(DEV), z.to(DEV))
        acts, off = TR.DEBUG["acts"], TR.DEBUG["off"]
    finally:
        TR.DEBUG = None
    R = raw_t.shape[1]
    names = TR.param_names(net, mode)
    p = _oracle_params(net, names)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]).double()
    # the saved pre-activations of the trunk against a float64 forward (nerfh_nff.py:547-553)
    e = O.freq_encode(pts.reshape(-1, 3), 10)
    h, M = e, N * S
    for l in range(1, 9):
        if l == 5:
            h = torch.cat([e, h], 1)
        pre = torch.nn.functional.linear(h, p[f"xyz_encoding_{l}.0.weight"].detach(), p[f"xyz_encoding_{l}.0.bias"].detach())
        got = acts[:, off[L.TB_L1 + l - 1]:off[L.TB_L1 + l - 1] + Wd, :].permute(0, 2, 1).reshape(-1, Wd)[:M].cpu().double()
        assert float((got - pre).abs().max()) < 5e-6, l
        h = torch.relu(pre)
    # ReLU is not differentiable at 0: an fp32 pre-activation within rounding of 0 may take the other branch than the
    # float64 oracle (seen: |pre| = 1.2e-9), which moves whole rank-1 terms of dW.  The oracle is therefore evaluated ON the
    # kernels' branch pattern (the masks the forward pass stored, tests/branch.py), after auditing that pattern against the
    # float64 pre-activations: every differing unit sits within fp32 rounding of zero.
    G = torch.randn(N, R, S, generator=g)
    (raw_t * G.to(DEV)).sum().backward()
    tag = "[h3]" if pipe == "h3" else ""
    assert set(ops.TIMERS) == {"field_fwd_train" + tag, "field_bwd_train" + tag}, set(ops.TIMERS)     # the pipe asked for ran
    pin = B.Pinned(tap, Wd)
    raw = O.query_field(p, pts, rays_d.double(), typ, typ == "fine", False, act=pin.act(True))          # [N,S,R]
    flips, units, worst_pre = pin.summary()
    P.record(f"train_field[{Wd},{C},{typ},{pipe}]", "relu branch flips vs float64", flips=flips, units=units, worst_preact_rel=worst_pre)
    assert worst_pre < 2e-5 and flips <= max(8, units // 100000), (flips, units, worst_pre)
    assert _relerr(raw_t.permute(0, 2, 1), raw) < 2e-5
    (raw * G.permute(0, 2, 1).double()).sum().backward()
    sd = dict(net.named_parameters())
    worst = ("", 0.)
    for n in names:
        assert sd[n].grad is not None, n
        e_ = _relerr(sd[n].grad, p[n].grad)
        worst = max(worst, (n, e_), key=lambda t: t[1])
    P.record(f"train_field[{Wd},{C},{typ},{pipe}]", "worst parameter gradient [branch-pinned]", e_hip=worst[1], e_ref=None, bound=1e-4)
    assert worst[1] < 1e-4, worst


@pytest.mark.parametrize("Wd,C,typ", [(128, 128, "coarse"), (128, 128, "fine"), (256, 16, "fine")])
def test_layered_dx_chain_matches_fused(Wd, C, typ, monkeypatch):
    """train.FUSED_DX = False (nefes_train_dx layer by layer: the fallback, e.g. for external embeddings) gives the gradients of
    the fused backward launch; same fp32 forward for both, so only the dX chain differs."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd import train as TR
    monkeypatch.setattr(ops, "SPLIT", "f32")
    torch.manual_seed(3)
    net = _net(typ, Wd, C)
    mode = L.FIELD_STATIC if typ == "coarse" else L.FIELD_FULL
    N, S = 37, 24
    o = torch.randn(N, 3, device=DEV) * 0.3
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    z = torch.sort(torch.rand(N, S, device=DEV) * 3.5 + 0.2, -1)[0]
    grads, G = {}, None
    for fused in (True, False):
        monkeypatch.setattr(TR, "FUSED_DX", fused)
        net.zero_grad()
        raw = TR.field_train(net, mode, o, d, d, z)
        G = torch.randn_like(raw) if G is None else G
        (raw * G).sum().backward()
        grads[fused] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert len(grads[True]) == len(grads[False]) >= 24
    for n, b in grads[True].items():
        assert _relerr(grads[False][n], b.cpu()) < 1e-5, n


@pytest.mark.parametrize("Wd,C", [(128, 128), (256, 16)])
def test_static_head_of_a_transient_network_on_both_pipes(Wd, C, monkeypatch):
    """NEFES_FIELD_STATIC on a network that HAS transient heads (a fine network evaluated with output_transient=False): the
    static-head fp16 streams of such a network share dir_encoding's scale with transient_encoding.0 (pack.cpp).  Raw outputs and
    every parameter gradient of the fp16 pipe against the fp32-MFMA pipe."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd import train as TR
    torch.manual_seed(8)
    net = _net("fine", Wd, C)
    N, S = 29, 20
    o = torch.randn(N, 3, device=DEV) * 0.3
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    z = torch.sort(torch.rand(N, S, device=DEV) * 3.5 + 0.2, -1)[0]
    res, G = {}, None
    for pipe in ("f32", "h3"):
        monkeypatch.setattr(ops, "SPLIT", pipe)
        monkeypatch.setattr(ops, "TIMERS", {})
        net.zero_grad()
        raw = TR.field_train(net, L.FIELD_STATIC, o, d, d, z)
        G = torch.randn_like(raw) if G is None else G
        (raw * G).sum().backward()
        assert ("field_fwd_train[h3]" in ops.TIMERS) == (pipe == "h3")
        res[pipe] = (raw.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    assert raw.shape[1] == 3 + C + 1 and len(res["h3"][1]) == 24
    assert _relerr(res["h3"][0], res["f32"][0].cpu()) < 2e-5
    for n, b in res["f32"][1].items():
        assert _relerr(res["h3"][1][n], b.cpu()) < 1e-4, n


@pytest.mark.parametrize("Wd,C,Ni", [(128, 128, 0), (256, 16, 32)])
def test_render_train_mode_weight_grads(Wd, C, Ni):
    """run_nefes.py-style step (test_time=False, trainable NeRF weights) through render(): loss on rgb, rgb0, features;
    Ni=0 is BASELINE configs[0]'s colour-only stage (coarse net, static head, compositing variant C)."""
    from nefes_amd.render import render
    H, W, focal, Nc = 8, 8, 12.0, 32
    coarse, fine = _net("coarse", Wd, C), _net("fine", Wd, C)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True)
    kw = dict(network_query_fn=None, perturb=0., N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    c2w = O.bench_pose()
    rays_o, rays_d = O.ray_bundle(H, W, focal, c2w)
    gen = torch.Generator().manual_seed(4)
    t_rgb, t_feat = torch.rand(H * W, 3, generator=gen), torch.randn(H * W, C, generator=gen)

    def loss_of(rgb, ex):
        l = ((rgb - t_rgb.to(rgb)) ** 2).mean() + ((ex["feat_map"] - t_feat.to(rgb)) ** 2).mean()
        if "rgb0" in ex and ex["rgb0"] is not None:
            l = l + ((ex["rgb0"] - t_rgb.to(rgb)) ** 2).mean()
        return l

    with B.tapped() as tap:
        rgb, disp, acc, ex = render(H, W, focal, rays=(rays_o.to(DEV), rays_d.to(DEV)), near=0., far=4., **kw)
    loss = loss_of(rgb, ex)
    loss.backward()
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, transient_at_test=True, NeRFW=True)
    from nefes_amd import lib as L
    from nefes_amd import train as TR
    pc = _oracle_params(coarse, TR.param_names(coarse, L.FIELD_STATIC))
    pf = _oracle_params(fine, TR.param_names(fine, L.FIELD_FULL))
    # the float64 oracle on the kernels' ReLU branch pattern (coarse pass = first tapped mask set, fine pass = second) and
    # at the kernels' sample depths: weight gradients comparable to 1e-4 instead of a kink-dominated 3e-2 (tests/branch.py)
    pin_c = B.Pinned(tap, Wd, 0)
    pin_f = B.Pinned(tap, Wd, 1) if Ni > 0 else None
    rgb_r, _, _, ex_r = O.render(H, W, focal, pc, pf, cfg, rays=(rays_o.double(), rays_d.double()), near=0., far=4.,
                                 coarse_act=pin_c.act(True), fine_act=None if pin_f is None else pin_f.act(True),
                                 z_fine=None if pin_f is None else pin_f.z_fine)
    loss_r = loss_of(rgb_r, ex_r)
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 1e-5 * abs(float(loss_r.detach()))
    tag = f"train_render[{Wd},{C},{Ni}]"
    for name, pin in (("coarse", pin_c), ("fine", pin_f)):
        if pin is not None:
            flips, units, worst_pre = pin.summary()
            P.record(tag, f"relu branch flips vs float64 ({name})", flips=flips, units=units, worst_preact_rel=worst_pre)
            assert worst_pre < 2e-5 and flips <= max(8, units // 100000), (name, flips, units, worst_pre)
    checked, worst = 0, ("", 0.)
    for net, p in ((coarse, pc), (fine, pf)):
        for n, t in net.named_parameters():
            if n in p and p[n].grad is not None and float(p[n].grad.abs().max()) > 0:
                assert t.grad is not None, n
                worst = max(worst, (n, _relerr(t.grad, p[n].grad)), key=lambda t_: t_[1])
                checked += 1
    P.record(tag, "worst parameter gradient [branch-pinned]", e_hip=worst[1], e_ref=None, bound=2e-4)
    assert worst[1] < 2e-4, worst
    assert checked >= 24


def test_training_steps_reduce_loss():
    """A few Adam steps of the stage-1 colour loss through render() (run_nefes.py:42-108): the packed weight streams are
    rebuilt after every optimiser step (NeRFH_NFF.packed() keys on the parameters' versions) and the loss goes down."""
    from nefes_amd.render import render
    H, W, focal = 16, 16, 24.0
    coarse = _net("coarse", 128, 128)
    prm = [p for n, p in coarse.named_parameters() if not n.startswith(("fusion_net", "exposure_embedding"))]
    opt = torch.optim.Adam(prm, lr=5e-4)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True)
    kw = dict(network_query_fn=None, perturb=1., N_importance=0, N_samples=32, network_fn=coarse, network_fine=None,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    ro, rd = O.ray_bundle(H, W, focal, O.bench_pose())
    ro, rd = ro.reshape(-1, 3).to(DEV), rd.reshape(-1, 3).to(DEV)
    target = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    losses = []
    for _ in range(8):
        rgb, _, _, ex = render(H, W, focal, rays=(ro, rd), near=0., far=4., **kw)
        loss = ((rgb - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.9 * losses[0], losses


@pytest.mark.parametrize("tag", ["stage1", "full"])
def test_train_mode_vs_reference_golden(golden, tag):
    """The HIP train path against tensors captured from the reference itself (tools/make_goldens.py, train.npz):
    maps and train extras within 2e-5; weight gradients with the kink-tolerant bound (see above)."""
    check_train_golden(golden("train"), tag)


def check_train_golden(g, tag):
    import numpy as np
    from nefes_amd.render import render
    Wd, C, Nc, Ni, H, W, focal = g[f"{tag}.cfg"]
    Wd, C, Nc, Ni, H, W = int(Wd), int(C), int(Nc), int(Ni), int(H), int(W)
    coarse, fine = _net("coarse", Wd, C), _net("fine", Wd, C)
    args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True)
    kw = dict(network_query_fn=None, perturb=0., N_importance=Ni, N_samples=Nc, network_fn=coarse, network_fine=fine,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
    rays_o, rays_d = O.ray_bundle(H, W, float(focal), torch.from_numpy(g[f"{tag}.c2w"])[:3, :4])
    rgb, disp, acc, ex = render(H, W, float(focal), rays=(rays_o.to(DEV), rays_d.to(DEV)), near=0., far=4., **kw)
    rel = lambda a, b: float(np.abs(a.detach().cpu().numpy() - b).max() / max(np.abs(b).max(), 1e-12))
    assert rel(rgb, g[f"{tag}.rgb"]) < 2e-5 and rel(acc, g[f"{tag}.acc"]) < 2e-5
    for k in [k for k in g if k.startswith(f"{tag}.ex.")]:
        assert rel(ex[k.split(".ex.")[1]], g[k]) < 5e-5, k
    t_rgb, t_feat = torch.from_numpy(g[f"{tag}.t_rgb"]).to(DEV), torch.from_numpy(g[f"{tag}.t_feat"]).to(DEV)
    loss = ((rgb - t_rgb) ** 2).mean() + ((ex["feat_map"] - t_feat) ** 2).mean()
    if Ni > 0:
        loss = loss + ((ex["rgb0"] - t_rgb) ** 2).mean()
    assert abs(float(loss.detach()) - float(g[f"{tag}.loss"])) < 1e-5 * float(g[f"{tag}.loss"])
    loss.backward()
    # the float64 oracle on ITS OWN branches (unpinned) for the three-way record: how far the reference's own fp32 gradients are
    # from float64 on this problem is what the 3e-2 below has to be read against (the branch-pinned twin above holds 2e-4)
    from nefes_amd import lib as L
    from nefes_amd import train as TR
    pc = _oracle_params(coarse, TR.param_names(coarse, L.FIELD_STATIC))
    pf = _oracle_params(fine, TR.param_names(fine, L.FIELD_FULL))
    cfg = O.RenderCfg(N_samples=Nc, N_importance=Ni, perturb=0., test_time=False, transient_at_test=True, NeRFW=True)
    rgb_r, _, _, ex_r = O.render(H, W, float(focal), pc, pf, cfg, rays=(rays_o.double(), rays_d.double()), near=0., far=4.)
    loss_r = ((rgb_r - t_rgb.cpu().double()) ** 2).mean() + ((ex_r["feat_map"] - t_feat.cpu().double()) ** 2).mean()
    if Ni > 0:
        loss_r = loss_r + ((ex_r["rgb0"] - t_rgb.cpu().double()) ** 2).mean()
    loss_r.backward()
    n = 0
    worst = {"e_hip": 0., "e_ref": 0., "direct": 0.}
    for k in [k for k in g if k.startswith(f"{tag}.grad.")]:
        net, name = k[len(f"{tag}.grad."):].split(".", 1)
        got = dict((coarse if net == "coarse" else fine).named_parameters())[name].grad
        assert got is not None, k
        a, b = got.detach().cpu().double().reshape(-1), torch.from_numpy(g[k]).double().reshape(-1)
        if float(b.abs().max()) == 0.:                                # e.g. transient_beta: beta is not in this loss
            assert float(a.abs().max()) == 0., k
            continue
        direct = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        t64 = (pc if net == "coarse" else pf)[name].grad
        if t64 is not None and float(t64.abs().max()) > 0:
            t64 = t64.reshape(-1)
            sc = t64.abs().max()
            worst["e_hip"] = max(worst["e_hip"], float((a - t64).abs().max() / sc))
            worst["e_ref"] = max(worst["e_ref"], float((b - t64).abs().max() / sc))
        worst["direct"] = max(worst["direct"], direct)
        assert direct < 1e-3, k                                       # against the reference's own fp32 gradient (measured <= 6e-5)
        assert float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)) > 0.9995, k
        n += 1
    # unpinned: both fp32 runs sit the same few 1e-3 from the float64 oracle on ITS branches (kink noise); the HIP run is held to the
    # shared rule against the reference's own distance, and to 1e-3 of the reference's gradient directly (above)
    bound = P.bound(worst["e_ref"], tol=1e-3)
    P.record(f"train_golden[{tag}]", "worst parameter gradient, UNPINNED: hip / reference fp32 vs float64 on its own branches", bound=bound, **worst)
    assert worst["e_hip"] <= bound, worst
    assert n >= 19


@pytest.mark.parametrize("Wd,C,typ,xyz", [(128, 128, "coarse", 63), (128, 128, "fine", 63), (256, 16, "fine", 63), (256, 16, "fine", 32)])
def test_device_repack_bit_identical(Wd, C, typ, xyz):
    """nefes_pack_device (one launch, from the concatenated parameters) == nefes_pack_weights (host) on the same values."""
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    torch.manual_seed(5)
    kw = dict(W=Wd, f_dim=C, in_channels_xyz=xyz)
    net = (NeRFH_NFF('coarse', **kw) if typ == "coarse" else NeRFH_NFF('fine', encode_appearance=True, encode_transient=True, **kw)).to(DEV)
    pk = net.packed()
    blob0 = pk.blob.clone()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))           # what an optimizer step does: same storage, new version
    pk2 = net.packed()
    assert pk2 is pk and pk.generation == 1              # re-packed in place, on the device
    assert not torch.equal(pk.blob, blob0)
    sd = {n: p for n, p in net.named_parameters()}
    host = ops.PackedField(sd, net.W, net.W_features, net.encode_transient, DEV, pk.xyz_encoding)
    # every stream the device packer produces is bit-identical to the host packer's -- fp32 fragments, bf16 triples, and the
    # fp16 two-part units with their scale tables (exponent per matrix, row bounds, bias maxima: csrc/pack_device.hip)
    assert pk.h3_valid and host.h3_valid
    if not torch.equal(pk.blob, host.blob):
        bad = (pk.blob != host.blob).nonzero().flatten()
        where = [k for k in range(13) if pk.info.stream[k].n_slabs and int(pk.info.stream[k].bias_off) <= int(bad[0])]
        raise AssertionError(f"{bad.numel()} bytes differ, first at {int(bad[0])} (stream {where[-1] if where else '?'})")
    assert len(pk.h3_byte_ranges()) >= 4
    assert net.packed() is pk and pk.generation == 1     # unchanged parameters: no work


def test_device_repack_without_fp16_plan_leaves_fp16_streams(monkeypatch):
    """ops.REPACK_H3 = False (NEFES_REPACK_H3=0): the round-2 behaviour -- fp16 streams untouched and flagged stale."""
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    monkeypatch.setattr(ops, "REPACK_H3", False)
    torch.manual_seed(5)
    net = NeRFH_NFF('coarse', W=128, f_dim=128).to(DEV)
    pk = net.packed()
    blob0 = pk.blob.clone()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))
    assert net.packed() is pk and not pk.h3_valid
    host = ops.PackedField({n: p for n, p in net.named_parameters()}, net.W, net.W_features, net.encode_transient, DEV, pk.xyz_encoding)
    same = torch.ones(pk.blob.numel(), dtype=torch.bool, device=DEV)
    for a, b in pk.h3_byte_ranges():
        same[a:b] = False
    assert torch.equal(pk.blob[same], host.blob[same])
    from nefes_amd import lib as L
    si = pk.info.stream[L.STREAM_FWD_SIGMA_H3]           # an all-fp16 stream: left exactly as the host packer wrote it
    a, b = int(si.slab_off), int(si.slab_off + si.n_slabs * 16 * 1024)
    assert torch.equal(pk.blob[a:b], blob0[a:b]) and not torch.equal(host.blob[a:b], blob0[a:b])


def test_backward_after_weight_update_raises():
    from nefes_amd import lib as L
    from nefes_amd.train import field_train
    torch.manual_seed(6)
    net = _net("coarse", 128, 128)
    N, S = 64, 32
    o, d = torch.randn(N, 3, device=DEV), torch.randn(N, 3, device=DEV)
    v = d / d.norm(dim=-1, keepdim=True)
    z = torch.linspace(0.1, 2.0, S, device=DEV).expand(N, S).contiguous()
    raw = field_train(net, L.FIELD_STATIC, o, d, v, z)
    with torch.no_grad():
        net.xyz_encoding_1[0].weight.mul_(1.5)
    net.packed()                                         # the next forward pass re-packs in place
    with pytest.raises(RuntimeError, match="modified"):
        raw.sum().backward()
