"""The N>1 launch path of bench.py under -m gpu.

`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a child
(bench.py: self_launch).  On the one-GPU test box the N ranks share cuda:0 and the pose-gradient all-reduce runs over gloo
(NEFES_BENCH_ONE_GPU=1 NEFES_BENCH_BACKEND=gloo: a functional check of the launch, the row shards and the collective, never a
measurement).  The file name sorts first among the GPU tests on purpose: every bench.py here is a fresh child process of a pytest
process that has not touched the GPU yet."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "1", "--height", "48", "--width", "64", "--cpu-rows", "0"]


def bench(extra, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=timeout, close_fds=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_through_the_self_launcher_match_one_rank():
    """--gpus 2 as a fresh child: torch.distributed.run starts two ranks, each renders its half of the rows, the 48-byte pose
    gradient is all-reduced; the JSON line says so and the reduced gradient equals the single-rank run's."""
    one_gpu = {"NEFES_BENCH_ONE_GPU": "1", "NEFES_BENCH_BACKEND": "gloo"}
    r1, j1 = bench(["--gpus", "1"] + SMALL, {})
    assert r1.returncode == 0 and j1 is not None, r1.stderr[-2000:]
    r2, j2 = bench(["--gpus", "2"] + SMALL, one_gpu)
    assert r2.returncode == 0 and j2 is not None, r2.stderr[-2000:]
    assert j1["n_gpus"] == 1 and j1["world_size"] == 1 and j1["collective_backend"] is None
    assert j2["n_gpus"] == 2 and j2["world_size"] == 2 and j2["collective_backend"] == "gloo"
    assert j2["config"]["parallelism"] == "rows/2" and j2["config"]["rays_per_step"] == 48 * 64
    assert j2["steps"] == 1 and j2["warmup"] == 1 and j2["value"] > 0 and j2["scaling"] == "strong"
    g1, g2 = j1["pose_grad_abs_max"], j2["pose_grad_abs_max"]
    assert g1 > 0 and abs(g1 - g2) <= 1e-6 * g1, (g1, g2)          # row shards + all-reduce == the whole frame on one rank


def test_launcher_started_ranks_must_match_gpus():
    """Started under a launcher with WORLD_SIZE != --gpus the bench refuses (it never under-runs silently)."""
    r, _ = bench(["--gpus", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_more_gpus_than_the_node_has_is_refused_with_exit_code_2():
    r, j = bench(["--gpus", "64"] + SMALL, {"NEFES_BENCH_ONE_GPU": "0"})
    assert r.returncode == 2 and j is None and "refusing" in r.stderr
