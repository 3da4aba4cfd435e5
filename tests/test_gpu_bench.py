"""GPU tests of bench.py itself on a test-sized workload: the JSON contract, the one-rank RCCL path
through the SAME child command `python bench.py --gpus N` launches for N > 1, and strong scaling's
bookkeeping."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAST = ["--workload", "tiny", "--steps", "2", "--warmup", "1", "--tunableop", "off", "--no-cpu-baseline"]


def _json_line(stdout):
    lines = [line for line in stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]          # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_line_contract_on_a_tiny_workload(hip_device):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + FAST + ["--extras", "off"],
                          env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-3000:]
    line = _json_line(done.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["workload"].startswith("tiny") and line["config"]["proposal"] == "tuned"
    assert line["value"] > 0 and line["rccl_world_size"] is None
    roofline = line["roofline"]
    assert roofline["bound"] == "hbm" and roofline["peak"] == 8000.0 and 0 < roofline["frac"] <= 1.0
    assert abs(line["value"] - 8 * 64 * 5 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_the_launcher_s_child_command_runs_on_one_rank_over_rccl(hip_device):
    """What `python bench.py --gpus N` starts for N > 1 — torch.distributed.run with one rank per GPU
    — run here with N = 1 and --force-dist: process group on RCCL, sharded loss path, one JSON line
    from rank 0 reporting the communicator's world size; weak scaling carries the strong-scaling
    block beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    argv = ["--gpus", "1", "--force-dist", "--extras", "off"] + FAST
    for attempt in range(2):
        done = subprocess.run(bench.child_command(argv, 1), env=bench.child_environment(env), capture_output=True,
                              text=True, timeout=900)
        if done.returncode == 0:
            break
    assert done.returncode == 0, done.stderr[-3000:]
    line = _json_line(done.stdout)
    assert line["rccl_world_size"] == 1 and line["n_gpus"] == 1
    assert line["config"]["global_batch"] == 8 and line["config"]["batch_per_gpu"] == 8
    assert "x1" in line["config"]["parallelism"] and line["value"] > 0
    strong = subprocess.run(bench.child_command(argv + ["--scaling", "strong"], 1), env=bench.child_environment(env),
                            capture_output=True, text=True, timeout=900)
    assert strong.returncode == 0, strong.stderr[-3000:]
    other = _json_line(strong.stdout)
    assert other["scaling"] == "strong" and other["config"]["global_batch"] == 8      # N = 1: the same job
    assert abs(other["loss"] - line["loss"]) < 1e-5 * abs(line["loss"])
