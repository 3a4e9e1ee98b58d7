"""Ray-batch data parallelism for the render-and-refine path: one process per GPU, rays sharded by
image rows, weights replicated (2.7 MB), and exactly one collective per backward: an all-reduce (SUM)
of the 3x4 pose gradient (48 bytes) over RCCL/xGMI.  The reference has no distributed code at all
(`--multi_gpu` is nn.DataParallel, nerfh_nff.py:647-648); this is new functionality.

No data-path collective exists in the forward pass: every rank generates its own rays from the
(replicated) pose.  If a whole-image loss follows (FusionNet), use `gather_maps`.
"""
import torch
import torch.distributed as dist


def row_shard(H: int, rank: int, world: int):
    """Contiguous block of image rows for `rank`: (row0, nrows).  Remainder rows go to the first ranks."""
    base, rem = divmod(H, world)
    nrows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, nrows


# Diagnostics of the one collective (bench.py sets this to a list for its timed region): per all-reduce a pair of events recorded on
# the current stream around the call (device time the stream spends in / waiting for the collective) and the host time of the call.
ALLREDUCE_TIMES = None
# True: a group of ONE rank still issues the all-reduce (an identity) -- how bench.py's NEFES_BENCH_FORCE_GROUP=1 takes the multi-rank
# code path through RCCL on a single-GPU box (tests/test_gpu_a_bench_launch.py)
ONE_RANK_COLLECTIVES = False


class _PoseGradAllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c2w, group):
        ctx.group = group
        return c2w.view_as(c2w)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(ctx.group) > 1 or ONE_RANK_COLLECTIVES):
            rec = ALLREDUCE_TIMES
            if rec is not None and g.is_cuda:
                import time
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
                host = time.perf_counter() - t0
                e1.record()
                rec.append((e0, e1, host))
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)      # the single collective: 12 floats
        return g, None


def replicate_pose(c2w: torch.Tensor, group=None) -> torch.Tensor:
    """Identity in forward; all-reduces the pose gradient in backward.  Wrap the pose once per iteration,
    render this rank's rows from the result, and every rank ends up with the full d loss / d c2w.

    The all-reduce is a SUM of the ranks' gradients: a per-rank loss must therefore be normalised by the size of the
    FULL frame (e.g. `(rgb**2).sum() / (H*W*3)`), not by the local row count -- a local `mean()` would weight the
    ranks' rays by world_size and mis-scale the pose gradient.  Every rank must run backward (the collective is in it)."""
    return _PoseGradAllReduce.apply(c2w, group)


def render_sharded(render_fn, H, W, focal, c2w, rank=None, world=None, group=None, **kw):
    """render() restricted to this rank's rows, with the pose-gradient all-reduce attached."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    row0, nrows = row_shard(H, rank, world)
    pose = replicate_pose(c2w, group)
    if nrows == 0:
        # more ranks than image rows: this rank renders nothing, but its (zero) loss must still reach the pose so that its
        # backward takes part in the all-reduce
        z = (pose.sum() * 0.).reshape(1, 1)
        C = getattr(kw.get("network_fn", None), "W_features", 0)
        return [z.expand(0, 3), z.expand(0, 1)[:, 0], z.expand(0, 1)[:, 0], {"feat_map": z.expand(0, C)}]
    return render_fn(H, W, focal, c2w=pose, row_range=(row0, nrows), **kw)


def gather_maps(local: torch.Tensor, H: int, group=None, W: int = None) -> torch.Tensor:
    """all_gather of per-rank [rows*W, C] maps into the full image order (for whole-image losses).
    Differentiable: the backward hands each rank the slice of the gradient that belongs to its rows.
    `W` (image width) is needed only when some rank may own zero rows (world_size > H)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    return _GatherRows.apply(local, H, group, W)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, H, group, W=None):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        mine = row_shard(H, rank, world)[1]
        if W is not None:
            per_row = int(W)
        elif mine > 0:
            per_row = local.shape[0] // mine
        else:
            raise ValueError("nefes_amd.dist.gather_maps: this rank owns no image rows (world_size > H); pass W=<image width>")
        sizes = [row_shard(H, r, world)[1] * per_row for r in range(world)]
        bufs = [local.new_empty((s,) + tuple(local.shape[1:])) for s in sizes]
        dist.all_gather(bufs, local.contiguous(), group=group) if len(set(sizes)) == 1 else \
            [dist.broadcast(b if r != rank else b.copy_(local), src=dist.get_global_rank(group, r) if group else r, group=group)
             for r, b in enumerate(bufs)]
        ctx.slice = (sum(sizes[:rank]), sizes[rank])
        return torch.cat(bufs, 0)

    @staticmethod
    def backward(ctx, g):
        off, n = ctx.slice
        return g[off:off + n].contiguous(), None, None, None
