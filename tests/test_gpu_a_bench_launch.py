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
# BASELINE configs[2] functionally: the frame's 480 rows over 8 ranks = 60 rows per rank, exactly the 8-GPU partition (the width is
# shrunk instead of the row count, so that eight ranks sharing one GPU stay a functional check)
ROWS480 = ["--steps", "1", "--warmup", "1", "--height", "480", "--width", "8", "--cpu-rows", "0"]


def bench(extra, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=timeout, close_fds=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("ranks,size", [(2, SMALL), (4, SMALL), (8, ROWS480)])
def test_ranks_through_the_self_launcher_match_one_rank(ranks, size):
    """--gpus N as a fresh child: torch.distributed.run starts N ranks, each renders its share of the rows, the 48-byte pose
    gradient is all-reduced; the JSON line says so and ALL TWELVE numbers of the reduced gradient equal the single-rank run's
    (fp32 sums in a different order: 1e-6 of the gradient's largest entry)."""
    one_gpu = {"NEFES_BENCH_ONE_GPU": "1", "NEFES_BENCH_BACKEND": "gloo"}
    SMALL = size
    n_rays = int(size[5]) * int(size[7])
    r1, j1 = bench(["--gpus", "1"] + SMALL, {})
    assert r1.returncode == 0 and j1 is not None, r1.stderr[-2000:]
    rn, jn = bench(["--gpus", str(ranks)] + SMALL, one_gpu)
    assert rn.returncode == 0 and jn is not None, rn.stderr[-2000:]
    assert j1["n_gpus"] == 1 and j1["world_size"] == 1 and j1["collective_backend"] is None
    assert jn["n_gpus"] == ranks and jn["world_size"] == ranks and jn["collective_backend"] == "gloo"
    assert jn["config"]["parallelism"] == f"rows/{ranks}" and jn["config"]["rays_per_step"] == n_rays
    assert jn["steps"] == 1 and jn["warmup"] == 1 and jn["value"] > 0 and jn["scaling"] == "strong"
    g1, gn = j1["pose_grad"], jn["pose_grad"]
    assert len(g1) == 12 and len(gn) == 12
    scale = max(abs(v) for v in g1)
    assert scale > 0 and abs(scale - j1["pose_grad_abs_max"]) <= 1e-12 * scale
    worst = max(abs(a - b) for a, b in zip(g1, gn))
    if worst > 1e-6 * scale:
        # N ranks on ONE GPU put torch's own loss kernels of one rank next to another rank's field kernels on the same CUs -- the
        # exposure of DESIGN.md 4.7 (a packed fp32 instruction with op_sel:[0,1] is wrong on lanes 48-63 next to another queue's
        # 16-bit MFMA kernel; this library carries none, torch's kernels are hipcc's to vectorise; a production launch has one rank per
        # GPU).  One wrong element of d loss / d rgb moves the pose gradient by ~1 / rays: far above 1e-6, far below a real error.
        # So: a miss is re-run once; the same miss twice is a failure, and so is anything above 1e-3.
        assert worst <= 1e-3 * scale, (g1, gn)
        rn, jn = bench(["--gpus", str(ranks)] + SMALL, one_gpu)
        assert rn.returncode == 0 and jn is not None, rn.stderr[-2000:]
        gn = jn["pose_grad"]
        worst = max(abs(a - b) for a, b in zip(g1, gn))
    assert worst <= 1e-6 * scale, (g1, gn)                         # row shards + all-reduce == the whole frame on one rank
    by_rank = jn["ms_per_step_by_rank"]
    assert len(by_rank["all"]) == ranks and 0 < by_rank["min"] <= by_rank["max"]
    assert abs(by_rank["max"] - jn["ms_per_step"]) <= 1e-6 * jn["ms_per_step"]      # the line's time is the slowest rank's
    # what a one-shot N-GPU run needs to explain itself: present in every N > 1 line
    m = jn["multi_gpu"]
    for key in ("all_reduce_ms_device_mean_by_rank", "all_reduce_ms_device_max_by_rank", "all_reduce_ms_host_call_mean_by_rank",
                "sustained_clock_ghz_by_rank", "sustained_16bit_mfma_tflops_by_rank"):
        assert len(m[key]) == ranks, (key, m[key])
    assert m["all_reduces_per_rank"] == 1 and all(t >= 0 for t in m["all_reduce_ms_device_mean_by_rank"])
    # (N ranks SHARE this box's one GPU: each rank's probe sees a fraction of the clock -- 0.4 ... 1.7 GHz -- ; one rank per GPU reads 1.6-1.7)
    assert all(0.0 < c < 3.0 for c in m["sustained_clock_ghz_by_rank"]), m["sustained_clock_ghz_by_rank"]
    assert m["collective_library"]["backend"] == "gloo" and "topology" in m
    assert "multi_gpu" not in j1
    # every line: the HBM-bound kernels against 8 TB/s and the whole step against the dominant kernel's bound
    for j in (j1, jn):
        hb = j["roofline"]["hbm_kernels"]
        # (N ranks sharing ONE GPU are time-sliced: a rank's kernel "duration" then includes other ranks' kernels and its GB/s can be
        # anything small -- only the fields and their upper bound are checked here)
        assert {"composite_fwd", "composite_bwd", "coarse_sample"} <= set(hb), hb
        assert all(v["ms"] > 0 and v["algorithmic_bytes_per_ray"] > 0 and 0 <= v["frac_of_8TB/s"] < 1.2 for v in hb.values()), hb
        assert 0 < j["roofline"]["end_to_end_frac"] < 1.0 and j["roofline"]["end_to_end_vs_fp32_mfma_peak"] > 0


def test_rccl_process_group_with_device_id_on_this_gpu():
    """The `nccl` (= RCCL) backend initialised the way bench.py initialises it on a multi-GPU node -- `device_id=` bound at
    init_process_group -- as a one-rank group on this box's GPU: init, an all-reduce, a barrier, and the 48-byte pose-gradient all-reduce
    in the backward of nefes_amd.dist.replicate_pose (tools/nccl_one_rank.py, run as a fresh child so that the collective library is
    loaded by a process that does nothing else).  What it cannot show: more than one rank over xGMI (no multi-GPU node in the pool)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_one_rank.py")], capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=600, close_fds=True)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("nccl 1-rank ok")]
    assert line, r.stdout[-1000:]
    _, _, _, ones, gsum = line[-1].split()
    assert float(ones) == 4.0                                  # all_reduce(SUM) of ones over one rank
    assert abs(float(gsum) - 2.0 * sum(range(12))) < 1e-4      # d/dc sum(c^2) = 2c through replicate_pose's all-reducing backward


def test_launcher_started_ranks_must_match_gpus():
    """Started under a launcher with WORLD_SIZE != --gpus the bench refuses (it never under-runs silently)."""
    r, _ = bench(["--gpus", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_more_gpus_than_the_node_has_is_refused_with_exit_code_2():
    r, j = bench(["--gpus", "64"] + SMALL, {"NEFES_BENCH_ONE_GPU": "0"})
    assert r.returncode == 2 and j is None and "refusing" in r.stderr


def test_the_n_rank_code_path_of_bench_through_rccl_under_the_drivers_launcher():
    """The driver's own N > 1 command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 1 and NEFES_BENCH_FORCE_GROUP=1: the process group is RCCL (`nccl`, bound to the
    device at init), and the run goes through every collective an 8-GPU run makes (barriers, the pose-gradient all-reduce inside
    backward with its events, the gathers of the per-rank times and diagnostics, the MAX of the step time).  The gradient of a group of
    one equals the plain run's bit for bit.  What stays unshown here: more than one rank over xGMI."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NEFES_BENCH_FORCE_GROUP="1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "NEFES_BENCH_BACKEND", "NEFES_BENCH_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--height", "48", "--width", "64", "--cpu-rows", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900, close_fds=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 1 and j["world_size"] == 1 and j["collective_backend"] == "nccl" and j["steps"] == 2
    m = j["multi_gpu"]
    assert m["all_reduces_per_rank"] == 2 and m["collective_library"]["backend"] == "nccl"
    assert len(m["collective_library"]["nccl_version"].split(".")) >= 2, m["collective_library"]
    assert any("GPU0" in ln or "SELF" in ln for ln in m["topology"]), m["topology"]          # the node's link table made it into the line
    assert len(m["all_reduce_ms_device_mean_by_rank"]) == 1 and m["all_reduce_ms_device_mean_by_rank"][0] >= 0, m
    assert 0.3 < m["sustained_clock_ghz_by_rank"][0] < 3.0, m               # one rank on its own GPU: the real sustained clock (1.55-1.71 seen)
    assert len(j["ms_per_step_by_rank"]["all"]) == 1
    r1, j1 = bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--height", "48", "--width", "64", "--cpu-rows", "0"], {})
    assert r1.returncode == 0 and j1 is not None, r1.stderr[-2000:]
    assert j1["pose_grad"] == j["pose_grad"]                                # all-reduce over one rank: the identity
