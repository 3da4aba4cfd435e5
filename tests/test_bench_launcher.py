"""CPU tests of bench.py's own launcher: `python bench.py --gpus N` must start N ranks itself (the
driver calls it without torchrun), before anything touches the GPU, and must also run unchanged as
a child of an external torchrun."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launcher_is_needed_only_outside_a_process_group():
    many = bench.parse(["--gpus", "8"])
    one = bench.parse([])
    assert one.gpus == 1 and one.workload == "c4" and one.proposal == "tuned" and one.scaling is None
    # the north-star curve is the headline of `--gpus N`: c4's B=1024 rows split over the ranks
    assert bench.resolve_scaling(many, 8) == "strong" and bench.resolve_scaling(one, 1) == "weak"
    assert bench.resolve_scaling(bench.parse(["--gpus", "8", "--workload", "c2"]), 8) == "weak"
    assert bench.resolve_scaling(bench.parse(["--gpus", "8", "--scaling", "weak"]), 8) == "weak"
    assert bench.needs_launcher(many, {}) is True
    assert bench.needs_launcher(many, {"WORLD_SIZE": "8", "RANK": "3"}) is False     # child of torchrun
    assert bench.needs_launcher(one, {}) is False


def test_child_command_and_environment():
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--scaling", "strong"]
    command = bench.child_command(argv, 4, port=23456, python="python3")
    assert command[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in command
    assert command[command.index("--nproc-per-node") + 1] == "4"
    assert command[command.index("--master-addr") + 1] == "127.0.0.1"      # the hostname may not resolve
    assert command[command.index("--master-port") + 1] == "23456"
    script = command.index(os.path.join(ROOT, "bench.py"))
    assert command[script + 1:] == argv                                     # flags reach every rank unchanged
    env = bench.child_environment({"PATH": "/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                         # dmabuf IPC for RCCL on this pool
    assert env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
    assert env["PATH"] == "/bin" and env["AESMC_BENCH_CHILD"] == "1"
    assert bench.child_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_dry_run_prints_the_launch_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                          env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr
    plan = json.loads(done.stdout.strip().splitlines()[-1])
    assert plan["launch"][1:3] == ["-m", "torch.distributed.run"] and "--dry-run" in plan["launch"]
    assert plan["scaling"] == "strong" and plan["workload"] == "c4"
    # the same command line as a child of torchrun: no second launch
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                          env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), capture_output=True, text=True,
                          timeout=300)
    assert done.returncode == 0, done.stderr
    plan = json.loads(done.stdout.strip().splitlines()[-1])
    assert plan == {"launch": None, "rank": 1, "world_size": 2, "workload": "c4", "scaling": "strong"}


def test_self_launch_runs_real_children_over_gloo_free_dry_run():
    """End to end through torch.distributed.run on this CPU box: the parent spawns two ranks, each
    parses the same flags, sees its RANK / WORLD_SIZE and (dry run) stops before the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for attempt in range(3):      # the probed rendezvous port can be taken before torchrun binds it
        command = bench.child_command(["--gpus", "2", "--dry-run", "--scaling", "strong"], 2)
        done = subprocess.run(command, env=bench.child_environment(env), capture_output=True, text=True, timeout=600)
        if done.returncode == 0:
            break
    assert done.returncode == 0, done.stderr[-2000:]
    # both ranks write to the one pipe they inherit: tolerate two objects landing on one line
    decoder, text, plans, at = json.JSONDecoder(), done.stdout, [], 0
    while True:
        at = text.find("{", at)
        if at < 0:
            break
        plan, at = decoder.raw_decode(text, at)
        plans.append(plan)
    assert sorted(p["rank"] for p in plans) == [0, 1], (done.stdout, done.stderr[-1000:])
    assert all(p["world_size"] == 2 and p["scaling"] == "strong" and p["launch"] is None for p in plans)
