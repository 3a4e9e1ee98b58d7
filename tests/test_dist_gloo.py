"""N>1 path on CPU: world_size-2 `gloo` processes exercise nefes_amd.dist (row sharding + the single
pose-gradient all-reduce + the row all-gather).  The renderer inside is the CPU oracle (tests may use it);
what is under test is the sharding arithmetic and the collective wiring, which are device-independent."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_render_rows(H, W, focal, c2w, row_range, pc, pf, cfg):
    from oracle import ref_cpu as O
    row0, n = row_range
    o, d = O.ray_bundle(H, W, focal, c2w)
    rgb, disp, acc, ex = O.render(H, W, focal, pc, pf, cfg, rays=(o[row0:row0 + n], d[row0:row0 + n]), near=0., far=4.)
    return rgb, ex["feat_map"]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nefes_amd import dist as D
    from oracle import ref_cpu as O
    H, W, focal, Wd, C = 6, 4, 3.5, 128, 128
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    cfg = O.RenderCfg(N_samples=16, N_importance=16)
    c2w = O.bench_pose().requires_grad_()
    row0, n = D.row_shard(H, rank, world)
    rgb, feat = _oracle_render_rows(H, W, focal, D.replicate_pose(c2w), (row0, n), pc, pf, cfg)
    # per-ray loss normalised by the FULL frame: the sum over ranks is the unsharded loss
    loss = (feat ** 2).sum() / (H * W * C) + (rgb ** 2).sum() / (H * W * 3)
    loss.backward()                       # <- the all-reduce of the 3x4 pose gradient happens here
    full = D.gather_maps(rgb.detach(), H)
    # by value (numpy): a torch tensor in a multiprocessing queue is handed over through the producer's shared-memory
    # descriptor, which is gone if this process exits before the parent has read it
    q.put((rank, c2w.grad.numpy().copy(), full.numpy().copy(), (row0, n)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_pose_gradient_equals_unsharded():
    from oracle import ref_cpu as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference
    H, W, focal, Wd, C = 6, 4, 3.5, 128, 128
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    cfg = O.RenderCfg(N_samples=16, N_importance=16)
    c2w = O.bench_pose().requires_grad_()
    rgb, feat = _oracle_render_rows(H, W, focal, c2w, (0, H), pc, pf, cfg)
    (O.bench_loss(rgb, feat)).backward()
    for rank, g, full, rows in got:
        g, full = torch.from_numpy(g), torch.from_numpy(full)
        assert torch.allclose(g, c2w.grad, rtol=1e-4, atol=1e-7), (rank, g, c2w.grad)   # identical on every rank
        assert torch.allclose(full, rgb.detach(), rtol=1e-5, atol=1e-6)                 # gathered image == unsharded image
    assert got[0][3] == (0, 3) and got[1][3] == (3, 3)


def test_row_shard_covers_every_row_once():
    from nefes_amd.dist import row_shard
    for H in (1, 7, 480, 481):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                row0, n = row_shard(H, r, world)
                rows += list(range(row0, row0 + n))
            assert rows == list(range(H))


# ---- a SHARDED refinement iteration (SURVEY.md section 8e's caveat; DFM_APR_refine.py:113-126): the fusion CNN and the feature loss need
# ---- the whole image, so the ranks' row shards of (rgb, feat) are gathered, every rank evaluates the replicated whole-image tail, and
# ---- gather_maps' backward hands each rank the slice of d loss / d maps that belongs to its rows; the pose gradients of the shards are
# ---- then summed by the one all-reduce.  world_size 2 and 3 (3 does not divide the 8 rows: unequal shards, broadcast path).
def _iteration_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nefes_amd import dist as D
    r = _sharded_iteration(D, rank, world)
    q.put((rank, r[0], r[1].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _iteration_problem():
    from nefes_amd.field import NeRFH_NFF
    from oracle import ref_cpu as O
    from oracle import refine_cpu as RC
    Wd, C, h, w, focal = 128, 16, 8, 6, 5.0
    pc, pf = O.make_field_params("coarse", Wd, C), O.make_field_params("fine", Wd, C)
    for p in (pc, pf):
        RC.structure_scene(p, 3.0, 1.0, 4.0)
    net = NeRFH_NFF('coarse', W=Wd, f_dim=C)                                     # seed-0 FusionNet / exposure parameters
    fsd = {k: v.detach().clone() for k, v in net.fusion_net.state_dict().items()}
    expo = net.exposure_embedding.params.detach().clone()
    gen = torch.Generator().manual_seed(3)
    target = torch.nn.functional.normalize(torch.randn(C, h, w, generator=gen), dim=0)
    hist = torch.full((1, 10), 10.)
    cfg = O.RenderCfg(N_samples=12, N_importance=12)
    return pc, pf, fsd, expo, target, hist, cfg, (h, w, focal, C)


def _sharded_iteration(D, rank, world):
    """One iteration's loss and pose gradient with the rows of the render sharded over `world` ranks (world = 1: unsharded)."""
    from oracle import ref_cpu as O
    from oracle import refine_cpu as RC
    pc, pf, fsd, expo, target, hist, cfg, (h, w, focal, C) = _iteration_problem()
    c2w = O.bench_pose().requires_grad_()
    pose = D.replicate_pose(c2w) if world > 1 else c2w
    row0, n = D.row_shard(h, rank, world)
    o, d = O.ray_bundle(h, w, focal, pose)
    rgb, _, _, ex = O.render(h, w, focal, pc, pf, cfg, rays=(o[row0:row0 + n], d[row0:row0 + n]), near=0., far=4., hist=hist)
    feat = ex["feat_map"]
    if world > 1:
        rgb, feat = D.gather_maps(rgb, h, W=w), D.gather_maps(feat, h, W=w)      # differentiable: backward = this rank's slice
    rgb = RC.affine_color_transform(expo, rgb, hist, 1)
    fused = RC.fusion_net(fsd, rgb, feat, h, w, 1)
    loss = RC.feature_loss(fused[0], target)
    loss.backward()                                                              # the pose-gradient all-reduce happens in here
    return float(loss.detach()), c2w.grad.detach().clone()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_refinement_iteration_equals_unsharded(world):
    from nefes_amd import dist as D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_iteration_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=480) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    loss1, g1 = _sharded_iteration(D, 0, 1)
    assert float(g1.abs().max()) > 0
    for rank, loss, g in got:
        assert abs(loss - loss1) <= 1e-6 * abs(loss1), (rank, loss, loss1)       # every rank holds the whole-image loss
        g = torch.from_numpy(g)
        assert float((g - g1).abs().max()) <= 2e-5 * float(g1.abs().max()), (rank, g, g1)      # and the full pose gradient
