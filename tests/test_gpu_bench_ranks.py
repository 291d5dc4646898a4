"""bench.py with MORE THAN ONE RANK on a one-GPU box: the ranks share the GPU and exchange over gloo (bench.py's test mode,
TRASE_BENCH_SHARED_GPU_TEST=1).  No multi-GPU node has been available to any round, so this is the only execution the N > 1 code
path of the bench gets before the driver's scaling run: launch line as the driver's, bucket + sink, the all-reduce per step, barrier +
MAX-over-ranks timing, pre-roll with collectives, rank 0 printing ONE JSON line last.  Not a measurement (the line says so)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_line_from_several_ranks_sharing_one_gpu(world):
    env = dict(os.environ, TRASE_BENCH_SHARED_GPU_TEST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--preroll-steps", "2", "--no-cpu-baseline", "--gaussians", "20000", "--width", "320", "--height", "192"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.strip()]
    d = json.loads(lines[-1])                              # the JSON line is the LAST thing on the job's stdout
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1, "more than one rank printed a line"
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert d["metric"].startswith("TEST MODE")
    assert d["value"] > 0 and abs(d["value"] - world * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]     # whole-job aggregate
    assert d["scaling"] == "weak"
    cfg = d["config"]
    assert cfg["bucket_bytes"] and cfg["exchange_algo"] == "allreduce" and cfg["exchange_ms"] is not None
    assert "view-DP" in cfg["workload"]


@pytest.mark.parametrize("extra,expect", [(["--exchange", "phased"], "phased"), (["--shard", "tiles"], "tiles"),
                                          (["--exchange", "direct", "--exchange-chunks", "1"], "direct")],
                         ids=["views-phased", "tiles", "views-direct"])
def test_eight_ranks_sharing_one_gpu_run_every_scaling_mode(extra, expect):
    """the first real 8-GPU run must not die on a code path that never executed.  Eight ranks of bench.py share
    this GPU over gloo -- the phased exchange (xyz first, the rest on a side stream until the next render), the direct exchange, and
    one view sharded by tile rows -- and rank 0 prints one well-formed line.  (Ranks sharing a GPU run their kernels side by side:
    correct since the library is built without packed-FP32 VALU, profiles/r6_two_streams.md.)  UNMEASURED on multi-GPU hardware."""
    world = 8
    env = dict(os.environ, TRASE_BENCH_SHARED_GPU_TEST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + len(expect)), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--preroll-steps", "1", "--no-cpu-baseline", "--gaussians", "12000", "--width", "320", "--height", "256"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1, "more than one rank printed a line"
    assert d["n_gpus"] == world and d["metric"].startswith("TEST MODE") and d["value"] > 0
    cfg = d["config"]
    if expect == "tiles":
        assert d["scaling"] == "strong" and "tile-row" in cfg["workload"]
    else:
        assert d["scaling"] == "weak" and cfg["bucket_bytes"]
        if expect == "phased":
            assert cfg["phased_bytes_first_rest"] and cfg["phased_bytes_first_rest"][0] > 0
