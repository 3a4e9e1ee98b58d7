"""Kernels of different HIP streams resident on the device at the same time (round 5: nefes_amd.refine.refine_concurrently).  Every
result must be the one the same launch sequence gives alone on the device, bit for bit: the kernels have no atomics and no shared
state, so anything else is a hazard.  The first version of this test found one: next to another queue's v_mfma_f32_32x32x16_f16
kernel, v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1] return a wrong low result on the wave's last 16 lanes -- hipcc's SLP vectoriser had
put one into composite_bwd4_kernel (and 97 more into 52 other kernels); the library is built without it now (DESIGN.md 4.7,
tools/store_hazard.py, tests/test_pack_stream.py)."""
import pytest
import torch

from tests.test_gpu_refine50 import T, nets, refiner

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_render_chain_is_bit_stable_next_to_another_streams_field_kernels():
    """field forward -> compositing -> loss -> compositing backward -> field backward -> ray reduction at the refinement frame's size
    (4 800 rays x 128 samples, 8 x 128 network, C = 128), alone and with the same chain / a bare field forward running on a second
    stream: every tensor of the chain equals its solo value."""
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    Wd, C, N, S = 128, 128, 4800, 128
    torch.manual_seed(0)
    fine = NeRFH_NFF('fine', W=Wd, f_dim=C, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    pk = fine.packed()
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    ro, rd = mk(N, 3) * 0.1, torch.nn.functional.normalize(mk(N, 3), dim=-1)
    z = (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(DEV)

    def chain():
        o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
        raw = ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL)
        early = []
        raw.register_hook(lambda g_: early.append(g_.clone()))
        rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, C, L.COMP_TRANSIENT, 0.03)
        ((rgb ** 2).sum() + (feat ** 2).sum() + (disp ** 2).sum()).backward()
        return dict(raw=raw.detach(), rgb=rgb.detach(), feat=feat.detach(), g_raw=early[0], g_rays=torch.cat([o.grad, d.grad, v.grad], 1))

    def field_forward():
        with torch.no_grad():
            return ops.FieldFromRays.apply(ro, rd, rd, z, pk, L.FIELD_FULL)

    solo = chain()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for other in (field_forward, chain):
        for rep in range(12):
            with torch.cuda.stream(streams[0]):
                a = chain()
            with torch.cuda.stream(streams[1]):
                b = other()
            torch.cuda.synchronize()
            for k, v in solo.items():
                assert torch.equal(a[k], v), (other.__name__, rep, k, int((a[k] != v).sum()))
            if isinstance(b, dict):
                for k, v in solo.items():
                    assert torch.equal(b[k], v), (other.__name__, rep, k, "second stream")


def test_two_images_on_two_streams_walk_their_solo_trajectories(golden):
    """refine_concurrently: two PoseRefiners (own buffers, own captured graph), two streams, 20 iterations -- poses and loss curves
    bit-identical to PoseRefiner.refine of each image alone (different starts, so the two graphs do different work)."""
    from nefes_amd.refine import refine_concurrently
    g = golden("refine50")
    refs = [refiner(g, graph=True), refiner(g, graph=True)]
    jobs = [(T(g["init_c2w"][k]), T(g["target_low"]), T(g["hist"])) for k in (0, 3)]
    n = 20
    solo = [r.refine(*job, iters=n) for r, job in zip(refs, jobs)]
    solo = [(p.clone(), l.clone()) for p, l in solo]
    for rep in range(3):
        outs = refine_concurrently(refs, jobs, iters=n)
        torch.cuda.synchronize()
        for (p, l), (ps, ls) in zip(outs, solo):
            assert torch.equal(p, ps) and torch.equal(l, ls), rep
    assert not torch.equal(solo[0][0], solo[1][0])
    with pytest.raises(ValueError):
        refine_concurrently([refs[0], refs[0]], jobs, iters=1)
    # more images than refiners: rounds of two, results in job order
    three = refine_concurrently(refs, [jobs[1], jobs[0], jobs[1]], iters=n)
    for (p, l), k in zip(three, (1, 0, 1)):
        assert torch.equal(p, solo[k][0]) and torch.equal(l, solo[k][1])


def test_refiners_that_share_a_fusion_net_do_not_touch_its_running_statistics(golden):
    """FusionNet's train-mode BatchNorm updates running_mean / running_var / num_batches_tracked with plain
    read-modify-writes; refiners that share the frozen networks would do that from several streams at once.  Such refiners are built
    with bn_running_stats=False (the outputs never depend on those buffers) and refine_concurrently refuses the other kind.  A solo
    refiner keeps the module's behaviour: the counter advances once per iteration it ran."""
    from nefes_amd.refine import refine_concurrently
    g = golden("refine50")
    shared = nets(g)
    bn = shared[0].fusion_net.net[-1]
    jobs = [(T(g["init_c2w"][k]), T(g["target_low"]), T(g["hist"])) for k in (0, 3)]
    n = 8
    tracking = [refiner(g, graph=True, networks=shared), refiner(g, graph=True, networks=shared)]
    with pytest.raises(ValueError, match="bn_running_stats=False"):
        refine_concurrently(tracking, jobs, iters=1)
    c0 = int(bn.num_batches_tracked)
    solo_pose, solo_loss = tracking[0].refine(*jobs[0], iters=n)
    torch.cuda.synchronize()
    ran = int(bn.num_batches_tracked) - c0
    assert ran >= n                                                   # (n iterations + whatever the first call's warm-up / capture ran)
    c1 = int(bn.num_batches_tracked)
    tracking[0].refine(*jobs[0], iters=n)
    torch.cuda.synchronize()
    assert int(bn.num_batches_tracked) - c1 == n                      # a captured graph replayed n times: exactly n updates
    quiet = [refiner(g, graph=True, networks=shared, bn_running_stats=False) for _ in range(2)]
    before = (int(bn.num_batches_tracked), bn.running_mean.clone(), bn.running_var.clone())
    outs = refine_concurrently(quiet, jobs, iters=n)
    torch.cuda.synchronize()
    assert int(bn.num_batches_tracked) == before[0] and torch.equal(bn.running_mean, before[1]) and torch.equal(bn.running_var, before[2])
    assert torch.equal(outs[0][0], solo_pose) and torch.equal(outs[0][1], solo_loss)      # same trajectory with or without the bookkeeping
    # a refiner on torch's glue is refused too (its kernels next to another stream's field kernels: DESIGN.md 4.7)
    with pytest.raises(ValueError, match="library's own kernels"):
        refine_concurrently([quiet[0], refiner(g, graph=False, networks=shared, bn_running_stats=False, fused_glue=False)], jobs, iters=1)


def test_two_images_of_the_default_mode_on_two_streams_equal_their_solo_runs(golden):
    """refine_apr_concurrently (`pose_only = 2`: the pose is a regression network's output, the loop trains a per-image copy of it): two
    PoseRefiner(pose_model=...) on two streams, 12 iterations, different regression networks (starts 0 and 3) -- pose, loss curve, PSNR /
    SSIM and the roll-back decision identical to refine_apr of each image alone; before that the regression network's own kernels are
    TRIED next to the library's field kernels (the empirical guard: they are torch's, exposed to DESIGN.md 4.7) and found bit-stable."""
    from nefes_amd.refine import _apr_kernels_bit_stable, refine_apr_concurrently
    from tests.test_gpu_refine50 import TinyAPR, photo_of, target_full
    g = golden("refine50")
    shared = nets(g)
    photo, tgt, hist = photo_of(g), target_full(g), T(g["hist"])
    n = 12
    refs = [refiner(g, graph=True, apr=TinyAPR(g["m2_weight"][k], g["m2_bias"][k]), networks=shared, bn_running_stats=False) for k in (0, 3)]
    jobs = [(photo, tgt, hist)] * 2
    solo = [r.refine_apr(*job, iters=n) for r, job in zip(refs, jobs)]
    solo = [(p.clone(), l.clone(), dict(i)) for p, l, i in solo]
    assert all(_apr_kernels_bit_stable(r, *job) for r, job in zip(refs, jobs))
    for rep in range(2):
        outs = refine_apr_concurrently(refs, jobs, iters=n)
        for (p, l, info), (ps, ls, infos) in zip(outs, solo):
            assert torch.equal(p, ps) and torch.equal(l, ls) and info == infos, rep
    assert not torch.equal(solo[0][0], solo[1][0])
    with pytest.raises(RuntimeError, match="pose_model"):
        refine_apr_concurrently([refiner(g, graph=True, networks=shared, bn_running_stats=False)], jobs[:1], iters=1)


@pytest.mark.parametrize("case", ["headline", "hashgrid", "train"])
def test_other_paths_are_bit_stable_next_to_another_streams_field_kernels(case):
    """The same question for the paths the refinement loop does not run: the headline network's chain (8 x 256, C = 16, 64 + 128 samples),
    the instances with the hash grid inside (configs[3]), and a train-mode step (activations saved by the forward, dX chain, dW kernels;
    outputs: the rendered colours and every parameter gradient)."""
    import types
    from nefes_amd import lib as L
    from nefes_amd import ops
    from nefes_amd.field import NeRFH_NFF
    from nefes_amd.render import render
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)

    def rays(N, S):
        return mk(N, 3) * 0.1, torch.nn.functional.normalize(mk(N, 3), dim=-1), (torch.rand(N, S, generator=g).sort(-1).values * 3 + 0.2).to(DEV)
    if case == "train":
        H = W = 96
        torch.manual_seed(0)
        coarse = NeRFH_NFF('coarse', W=128, f_dim=128).to(DEV)
        prm = [p for n, p in coarse.named_parameters() if not n.startswith(("fusion_net", "exposure_embedding"))]
        args = types.SimpleNamespace(nerfh_nff=True, use_fine_only=False, NeRFW=True, transient_at_test=True, netchunk=1024)
        kw = dict(network_query_fn=None, perturb=0., N_importance=0, N_samples=64, network_fn=coarse, network_fine=None, use_viewdirs=True,
                  white_bkgd=False, raw_noise_std=0., test_time=False, args=args, ndc=False, lindisp=False)
        ro, rd, _ = ops.raygen_fwd(H, W, 200.0, O.bench_pose().to(DEV))
        target = torch.rand(H * W, 3, generator=g).to(DEV)

        def fn():
            rgb, _, _, _ = render(H, W, 200.0, rays=(ro, rd), near=0., far=4., **kw)
            grads = torch.autograd.grad(((rgb - target) ** 2).mean(), prm)
            return torch.cat([rgb.detach().reshape(-1)] + [g_.reshape(-1) for g_ in grads])
    else:
        grid = None
        if case == "hashgrid":
            grid = ops.HashGrid(25.0, device=DEV)
            grid.table.mul_(3e3)
        fine = NeRFH_NFF('fine', W=256, f_dim=16, in_channels_xyz=32 if grid is not None else 63, encode_appearance=True,
                         encode_transient=True).requires_grad_(False).to(DEV)
        pk = fine.packed()
        ro, rd, z = rays(3200, 192)

        def fn():
            o, d, v = ro.clone().requires_grad_(), rd.clone().requires_grad_(), rd.clone().requires_grad_()
            raw = (ops.FieldFromRaysHashGrid.apply(o, d, v, z, pk, L.FIELD_FULL, grid) if grid is not None
                   else ops.FieldFromRays.apply(o, d, v, z, pk, L.FIELD_FULL))
            rgb, feat, disp, acc, depth, weights, beta = ops.Composite.apply(raw, z, 16, L.COMP_TRANSIENT, 0.03)
            ((rgb ** 2).sum() + (feat ** 2).sum() + (disp ** 2).sum()).backward()
            return torch.cat([rgb.detach().reshape(-1), feat.detach().reshape(-1), o.grad.reshape(-1), d.grad.reshape(-1), v.grad.reshape(-1)])
    nfine = NeRFH_NFF('fine', W=128, f_dim=128, encode_appearance=True, encode_transient=True).requires_grad_(False).to(DEV)
    npk = nfine.packed()
    nro, nrd, nz = rays(4800, 128)

    def neighbour():
        with torch.no_grad():
            return ops.FieldFromRays.apply(nro, nrd, nrd, nz, npk, L.FIELD_FULL)
    solo = fn().clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for other in (neighbour, fn):
        for rep in range(6):
            with torch.cuda.stream(streams[0]):
                a = fn()
            with torch.cuda.stream(streams[1]):
                b = other()
            torch.cuda.synchronize()
            assert torch.equal(a, solo), (case, other.__name__, rep, int((a != solo).sum()))
