"""Throughput of the SMC ELBO hot path on MI355X (BASELINE.json metric: particle-steps/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c5|c3|c4nl]

One "step" = one SMC ELBO evaluation (aesmc_amd.losses.get_loss(..., 'aesmc') forward, autograd
graph recorded as in training) over one synthetic batch already resident in HBM.  N > 1: the driver
launches one process per GPU via torch.distributed.run; each rank owns `B` batch rows (weak
scaling), the only collective is the all-reduce of sum_b log Z_b (RCCL over xGMI).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the resampling kernel (fused step K2+K3, or K3 alone), timed per launch with HIP events on its stream
                 while the same steps run again; achieved = algorithmic bytes / launch time;
  cpu_baseline : oracle/reference_port.py (the op-for-op CPU port of the reference, kind "port")
                 timed on this box's host cores on a bounded sample of the same workload;
  kernels      : the same per-launch figures for every kernel of the path.
"""
import argparse
import contextlib
import json
import os
import sys
import time

# before the HIP runtime starts: ROCm 7.0's graph fast path misorders captured memset nodes
# (aesmc_amd/__init__.py sets the same default on import; stated here because it shapes the numbers)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

FORWARD_ONLY = {"c4", "c5"}  # autograd retention of T x (dozens of [B,K,d] temporaries) would exceed HBM

ALGORITHM = {"c3": "iwae"}  # every other workload is the SMC ELBO ('aesmc')

# name: (description, model kind, d, per-GPU B, K, T)
WORKLOADS = {
    "c2": ("LGSSM d=10 B=256 K=1024 T=50, SMC ELBO (configs[1])", "lgssm", 10, 256, 1024, 50),
    "c3": ("one-step Gaussian IWAE d=1 B=4096 K=8192 T=1, no resampling (configs[2])", "gaussian", 1, 4096, 8192, 1),
    "c4": ("LGSSM d=10 B=1024 K=4096 T=100, SMC ELBO (north-star target shape)", "lgssm", 10, 1024, 4096, 100),
    "c4s": ("LGSSM d=10 B=128 K=4096 T=100, SMC ELBO (one GPU's shard of c4 at 8 GPUs)", "lgssm", 10, 128, 4096, 100),
    "c4nl": ("nonlinear SSM + MLP proposal d=10 B=128 K=4096 T=100 (configs[3] per-GPU shard)", "nonlinear", 10, 128, 4096, 100),
    "c4ls": ("nonlinear SSM, proposal net outputs loc and scale, learned vector transition scale, d=10 B=128 K=4096 T=100",
             "learned_scale", 10, 128, 4096, 100),
    "c5": ("LGSSM d=128 B=64 K=16384 T=200, SMC ELBO forward, degeneracy stress (configs[4])", "lgssm", 128, 64, 16384, 200),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--tunableop", default="on", choices=["on", "off"],
                    help="PyTorch TunableOp for the user callables' matmuls (tunes each GEMM shape once, in warm-up)")
    ap.add_argument("--tunableop-file", default=None,
                    help="keep TunableOp's picks in this CSV: a later run that finds it skips the tuning trials "
                         "(used to profile without them)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and take the sharded code path even with one rank (test hook)")
    ap.add_argument("--mode", default=None, choices=["graph", "eager"],
                    help="graph: replay the whole ELBO as one hipGraph (aesmc_amd.graphs); eager: Python loop")
    return ap.parse_args()


def build_model(kind, dim, device, state):
    from aesmc_amd.testing import models
    cls = {"lgssm": models.LgssmNd, "nonlinear": models.NonlinearSsm, "gaussian": models.GaussianIwae,
           "learned_scale": models.LearnedScaleSsm}[kind]
    if kind == "gaussian":
        return cls(state=state, validate_args=False).to(device)
    # validate_args=False: no per-call host sync inside torch.distributions (standard practice)
    return cls(dim, seed=0, state=state, validate_args=False).to(device)


def cpu_baseline(kind, dim, B, K, T, algorithm="aesmc", budget_s=20.0):
    """The CPU port of the reference (oracle/reference_port.py) on a BOUNDED sample of the same
    workload: same model, K and d; batch rows (and, if still too slow, timesteps) are cut until a
    calibrated estimate fits `budget_s` seconds.  Threads: min(cores, 16) — PyTorch's default of
    one thread per core is far slower on big hosts for these small ops."""
    from oracle import reference_port
    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    model = build_model(kind, dim, torch.device("cpu"), reference_port)
    parts = (model.initial, model.transition, model.emission, model.proposal)

    def run(b, t):
        observations = model.simulate(t, b, seed=1)
        np.random.seed(0)
        torch.manual_seed(0)
        t0 = time.perf_counter()
        loss = reference_port.get_loss(observations, K, algorithm, *parts)
        return time.perf_counter() - t0, float(loss.detach())

    def cost(b, t):  # per-step work + O(T^2) history re-gather
        return b * K * dim * (t + 0.35 * t * t)

    cal_b, cal_t = max(1, min(B, 32)), min(T, 6)
    run(cal_b, cal_t)                       # warm-up (thread pool, allocator)
    cal_s, _ = run(cal_b, cal_t)
    rate = cost(cal_b, cal_t) / max(cal_s, 1e-6)
    b, t = B, T
    while cost(b, t) / rate > budget_s and b > 8:
        b //= 2
    while cost(b, t) / rate > budget_s and t > 10:
        t -= 5
    dt, loss = run(b, t)
    return {"value": b * K * t / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": "1 forward ELBO, B={} K={} T={} d={} in {:.1f} s on {} of {} host cores; "
                      "oracle/reference_port.py (PyTorch-CPU + NumPy, keeps the reference's O(T^2) "
                      "history re-gather and per-row np.digitize loop)".format(b, K, t, dim, dt, threads, cores),
            "loss": loss}


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when the first communicator is created; the contract
    is ONE JSON line on stdout, so the banner is steered to stderr (file-descriptor level)."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus {} must be launched with torch.distributed.run "
                             "--nproc-per-node {}".format(args.gpus, args.gpus))
    import torch.distributed as dist
    import aesmc_amd
    from aesmc_amd import _kernels, distributed

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)          # creates the communicator now (and prints its banner)
            torch.cuda.synchronize()

    description, kind, dim, B, K, T = WORKLOADS[args.workload]
    algorithm = ALGORITHM.get(args.workload, "aesmc")
    provider = _kernels.get()
    assert provider.name == "hip"
    model = build_model(kind, dim, device, aesmc_amd.state)
    global_B = B * world
    observations = model.simulate(T, global_B, seed=1)          # same data on every rank ...
    observations = distributed.shard_observations(observations, rank, world)  # ... own rows only
    parts = (model.initial, model.transition, model.emission, model.proposal)
    np.random.seed(0)
    torch.manual_seed(0)
    if args.tunableop == "on":
        # hipBLASLt's default pick for the callables' [B*K, d] x [d, d] maps runs at ~1.3 TB/s;
        # TunableOp (a stock PyTorch feature) times the candidates on first use and keeps the best.
        try:
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
            torch.cuda.tunable.set_filename(args.tunableop_file or os.path.join(
                os.environ.get("TMPDIR", "/tmp"), "aesmc_tunableop_%d.csv" % os.getpid()))
            torch.cuda.tunable.set_max_tuning_duration(30)
            torch.cuda.tunable.set_max_tuning_iterations(20)
        except Exception as error:     # an optional PyTorch knob: the bench must not depend on it
            print("TunableOp unavailable ({}): running with PyTorch's default GEMM picks".format(error),
                  file=sys.stderr)
            args.tunableop = "off"

    if args.mode is None:  # the big forward-only shapes are device-bound and need the HBM for data
        args.mode = "eager" if args.workload in FORWARD_ONLY else "graph"
    grad_mode = torch.no_grad if args.workload in FORWARD_ONLY else torch.enable_grad
    if args.workload in FORWARD_ONLY:
        args.no_backward = True

    def step(backward=False):
        with grad_mode():
            return _step(backward)

    def _step(backward=False):
        if use_dist:
            loss = distributed.sharded_get_loss(observations, K, algorithm, *parts, global_batch_size=global_B,
                                                rank=rank, world_size=world)
        else:
            loss = aesmc_amd.losses.get_loss(observations, K, algorithm, *parts)
        if backward:
            model.zero_grad(set_to_none=True)
            loss.backward()
            if use_dist:
                distributed.all_reduce_gradients(list(model.parameters()))
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        """n calls of fn between barrier + synchronize on both sides; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = fn()
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), float(loss.detach())

    # ---- the step: one hipGraph replay of the whole ELBO (default) or the eager Python loop ------
    shard = (global_B, rank, world) if use_dist else None
    mode, graph_error, graphed = "eager", None, None
    if args.mode == "graph":
        try:
            from aesmc_amd import graphs
            with grad_mode():
                # check_flags=False: replays run back to back; the device status word (NaN weights,
                # degenerate rows, ...) is read once after each timed region instead of every step
                graphed = graphs.GraphedLoss(observations, K, algorithm, *parts, shard=shard, check_flags=False)
            mode = "hipgraph"
        except Exception as error:  # capture is an optimisation: report and fall back to eager
            graph_error = "{}: {}".format(type(error).__name__, error)
            torch.cuda.synchronize()
    forward = graphed if graphed is not None else step

    for _ in range(args.warmup):
        forward()
    seconds, loss = timed(forward, args.steps)
    if graphed is not None:
        graphed.check()
    ms_per_step = 1e3 * seconds / args.steps
    value = global_B * K * T * args.steps / seconds

    eager_value = None
    if graphed is not None:  # the eager loop beside it, for the record
        step()
        n = max(1, args.steps // 2)
        se, _ = timed(step, n)
        eager_value = global_B * K * T * n / se

    fwd_bwd = None
    if not args.no_backward:
        n = max(1, args.steps // 2)
        train_step = graphed_train = None
        if graphed is not None:
            try:
                from aesmc_amd import graphs
                graphed_train = graphs.GraphedLoss(observations, K, algorithm, *parts, backward=True, shard=shard,
                                                   check_flags=False)
                params = list(model.parameters())

                def train_step():
                    out = graphed_train()
                    if use_dist:
                        distributed.all_reduce_gradients(params)
                    return out
            except Exception as error:
                graph_error = "backward capture: {}: {}".format(type(error).__name__, error)
                torch.cuda.synchronize()
                train_step = graphed_train = None
        if train_step is None:
            def train_step():
                return step(backward=True)
        train_step()
        sb, _ = timed(train_step, n)
        if graphed_train is not None and train_step is not None:
            graphed_train.check()
        fwd_bwd = global_B * K * T * n / sb

    # ---- per-kernel timing: the same steps again with HIP events around every launch ------------
    provider.timer = _kernels.KernelTimer()
    for _ in range(args.steps):
        step()
    kernels = provider.timer.summary()
    provider.timer = None
    # the roofline kernel: the resample gather (K3); a workload that never resamples (c3, IWAE) is
    # what BASELINE.json uses to isolate the fused log-weight + log-sum-exp kernel (K1)
    if "moved_GBps" in kernels.get("resample_step", {}):
        name, label = "resample_step", "ancestor_index_inv_kernel with payload (fused step: K2 + K3)"
    elif "resample_gather" in kernels:
        name, label = "resample_gather", "resample_gather_kernel (K3)"
    else:
        name, label = "logweight_lse", "logweight_lse_kernel (K1)"
    dominant = kernels.get(name)
    roofline = None
    if dominant:
        achieved = dominant["GBps"]
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):  # PMC passes are separate rocprofv3 runs (tools/pmc_gather.py)
            with open(pmc_path) as fh:
                traffic = json.load(fh).get(args.workload, {}).get(name + "_bytes_per_launch")
        roofline = {"kernel": label, "bound": "hbm", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": traffic, "avg_launch_us": round(dominant["avg_us"], 2),
                    "algorithmic_bytes_per_launch": dominant["bytes_per_launch"],
                    "launches": dominant["launches"]}
        if "unique_ancestor_fraction" in dominant:
            # `achieved` prices a full read of the source; with a collapsed particle
            # system only the surviving rows are fetched, so it can exceed what HBM moved: the
            # surviving fraction and the bytes that did move on these operands are stated beside it.
            roofline["unique_ancestor_fraction"] = round(dominant["unique_ancestor_fraction"], 4)
            roofline["achieved_moved_bytes"] = round(dominant["moved_GBps"], 1)
            roofline["frac_moved_bytes"] = round(dominant["moved_GBps"] / HBM_PEAK_GBPS, 4)

    out = {
        "metric": "particle_steps_per_sec", "value": value, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "{}: {}".format(args.workload, description), "batch_per_gpu": B,
                   "global_batch": global_B, "num_particles": K, "num_timesteps": T, "state_dim": dim,
                   "parallelism": "batch-shard x{} (one RCCL all-reduce of sum log Z per ELBO)".format(world),
                   "pytorch": "TunableOp {} for the user callables' matmuls; distributions built with "
                              "validate_args=False".format(args.tunableop),
                   "hip_runtime": "DEBUG_CLR_GRAPH_PACKET_CAPTURE={} (0: captured memset nodes keep stream "
                                  "order on ROCm 7.0)".format(os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")),
                   "step": "one forward ELBO, get_loss(..., '{}'), ".format(algorithm) +
                           ("torch.no_grad()" if args.workload in FORWARD_ONLY else "autograd graph recorded") +
                           (", all T timesteps replayed as one hipGraph" if mode == "hipgraph" else ", eager Python loop")},
        "loss": loss,
        "mode": mode, "graph_error": graph_error, "tunableop": args.tunableop,
        "eager_particle_steps_per_sec": eager_value,
        "fwd_bwd_particle_steps_per_sec": fwd_bwd,
        "roofline": roofline,
        "kernels": {k: {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in kernels.items()},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, dim, B, K, T, algorithm)
    if use_dist:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
