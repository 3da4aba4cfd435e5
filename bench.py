"""Throughput of the SMC ELBO hot path on MI355X (BASELINE.json metric: particle-steps/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|c4s|c4nl|c4ls|c5]
                    [--proposal tuned|stock] [--scaling weak|strong] [--extras on|off]

One "step" = one SMC ELBO evaluation (aesmc_amd.losses.get_loss(..., 'aesmc') forward, autograd
graph recorded as in training) over one synthetic batch already resident in HBM.  The default
workload is the north-star shape of BASELINE.json: LGSSM d=10, B=1024, K=4096, T=100 ("c4").
Up to 2^22 particles per timestep (c4 included) the ELBO is captured once and replayed as one hipGraph
(aesmc_amd.graphs.GraphedLoss: fresh uniforms and noise per replay, verified against eager evaluations); `mode` says
so and `eager_particle_steps_per_sec` gives the plain Python loop's figure beside it (`--mode eager` times that).

N > 1: `python bench.py --gpus N` starts N fresh children itself (python -m
torch.distributed.run, one rank per GPU, RCCL) before anything touches the GPU; it also runs as a
child of an external torchrun (RANK / WORLD_SIZE in the environment).  Batch rows shard over the
ranks; the only collective on the data path is the all-reduce of sum_b log Z_b.  `--scaling strong`
(the default for c4, BASELINE.json's north-star shape): the workload's B rows are split over the ranks;
`--scaling weak` (the default elsewhere): every rank owns the workload's B rows.  The other curve rides in
`extras`, the all-reduce's own time in `allreduce_us_per_elbo`.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel of the timed region (K16 — the launch that holds the resampling gather — on the
                 linear-Gaussian route; the fused step K2+K3 or K3 elsewhere; K1 for the IWAE workload), timed per
                 launch with HIP events on its stream on operands sampled from the timed steps; achieved =
                 algorithmic bytes / time, next to the PMC traffic and the bytes that had to move given how many
                 ancestors survived;
  cpu_baseline : oracle/reference_port.py (the op-for-op CPU port of the reference, kind "port")
                 timed on this box's host cores on a bounded sample of the same workload;
  kernels      : the same per-launch figures for every kernel of the path;
  extras       : (N = 1) the same workload with SURVEY.md 8(d)'s untrained proposal stand-in,
                 configs[1] replayed as one hipGraph (round 1's headline), kernel legs at the
                 shapes BASELINE.json uses to isolate K1 and K3, and the float32 index flip rate
                 against the reference's fixtures.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

# before the HIP runtime starts: ROCm 7.0's graph fast path misorders captured memset nodes
# (aesmc_amd/__init__.py sets the same default on import; stated here because it shapes the numbers)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_FP32_PEAK_TFLOPS = 157.3  # dense fp32 matrix-core peak (v_mfma_f32_16x16x4_f32; MI355X_MICROARCH.md: 155 TF measured)

# name: description, model kind, d, B (per GPU under weak scaling), K, T, model keyword arguments
WORKLOADS = {
    "c4": ("LGSSM d=10 B=1024 K=4096 T=100, SMC ELBO (north-star target shape)", "lgssm", 10, 1024, 4096, 100, {}),
    "c2": ("LGSSM d=10 B=256 K=1024 T=50, SMC ELBO (configs[1])", "lgssm", 10, 256, 1024, 50, {}),
    "c3": ("one-step Gaussian IWAE d=1 B=4096 K=8192 T=1, no resampling (configs[2])", "gaussian", 1, 4096, 8192, 1, {}),
    "c4s": ("LGSSM d=10 B=128 K=4096 T=100, SMC ELBO (one GPU's shard of c4 at 8 GPUs)", "lgssm", 10, 128, 4096, 100, {}),
    "c4x2": ("LGSSM d=10 B=512 K=4096 T=100, SMC ELBO (one GPU's shard of c4 at 2 GPUs, strong scaling)", "lgssm", 10, 512, 4096, 100, {}),
    "c4x4": ("LGSSM d=10 B=256 K=4096 T=100, SMC ELBO (one GPU's shard of c4 at 4 GPUs, strong scaling)", "lgssm", 10, 256, 4096, 100, {}),
    "c4nl": ("nonlinear SSM + MLP proposal d=10 B=128 K=4096 T=100 (configs[3] per-GPU shard)", "nonlinear", 10, 128, 4096, 100, {}),
    "c4ls": ("nonlinear SSM, proposal net outputs loc and scale, learned vector transition scale, d=10 B=128 K=4096 T=100",
             "learned_scale", 10, 128, 4096, 100, {}),
    "c5": ("LGSSM d=128 B=64 K=16384 T=200, SMC ELBO forward, degeneracy stress (configs[4])", "lgssm", 128, 64, 16384, 200, {}),
    # configs[4]'s shape on a particle system that is NOT collapsed: at d=128 even the locally optimal
    # proposal keeps only ~20 % of the ancestors per step with SURVEY's emission noise 0.5; with 0.05
    # (an informative sensor) half of them survive, so the gather reads what it is priced for
    "c5h": ("LGSSM d=128 B=64 K=16384 T=200, emission scale 0.05, SMC ELBO forward (configs[4] shape, healthy particle system)",
            "lgssm", 128, 64, 16384, 200, {"emission_scale": 0.05}),
}
for _rows in (320, 384, 448, 640, 768):      # intermediate shards of the north-star shape (graph-replay / host-bound studies)
    WORKLOADS["c4b%d" % _rows] = ("LGSSM d=10 B=%d K=4096 T=100, SMC ELBO (a part of c4's batch)" % _rows, "lgssm", 10,
                                  _rows, 4096, 100, {})
# the reference's own model classes, untouched (test/models/lgssm.py: Python-number scales, default validate_args, the
# cat / view proposal), at configs[1]'s B, K, T: what a user who switches packages runs on day one
WORKLOADS["c2ref"] = ("1-D LGSSM written as the reference writes it (test/models/lgssm.py classes) B=256 K=1024 T=50, SMC ELBO",
                      "lgssm1d_reference", 1, 256, 1024, 50, {})
WORKLOADS["tiny"] = ("LGSSM d=3 B=8 K=64 T=5 (test-sized: exercises every leg of this file in seconds)",
                     "lgssm", 3, 8, 64, 5, {})
ALGORITHM = {"c3": "iwae"}          # every other workload is the SMC ELBO ('aesmc')
NO_GRAD = {"c5", "c5h"}             # autograd retention of T x [B,K,128] temporaries exceeds HBM: forward under no_grad
# B*K at or below this: the ELBO is captured once and replayed as one hipGraph (aesmc_amd.graphs.GraphedLoss).  Small
# shapes are host-bound in the eager loop; at the north-star shape (2^22 particles) the replay still runs 3-5 % ahead of
# the eager loop (13.7-14.0 against 14.2-14.6 ms in five sessions of round 4) — the eager figure is printed beside it
GRAPH_PARTICLES = 1 << 22


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--proposal", default="tuned", choices=["tuned", "stock"],
                    help="LGSSM workloads: 'tuned' = the model's locally optimal proposal in closed form (what "
                         "training converges towards: a healthy particle system); 'stock' = SURVEY.md 8(d)'s "
                         "untrained linear stand-in (collapses to ~14 %% surviving ancestors per step)")
    ap.add_argument("--callables", default="affine", choices=["affine", "matmul"],
                    help="LGSSM workloads: how the model's callables state their linear-Gaussian terms. 'affine': "
                         "aesmc_amd.linear_gaussian.AffineNormal(source, weight, scale, offset) — the locations are "
                         "evaluated inside the sampling / weighting kernels, the proposal's draw deferred to the launch that "
                         "weighs it (K15; a step's backward: K14); 'matmul': "
                         "Normal(source @ weight.T + offset, scale) with PyTorch matmuls, as round 1 and 2 timed it")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1 only. strong: the workload's B rows are split over the ranks (value = B K T / wall: the "
                         "north-star curve of c4, B=1024 over 8 GPUs); weak: every rank owns B rows.  Default: strong for "
                         "c4 — the shape BASELINE.json's scaling target names — weak elsewhere")
    ap.add_argument("--extras", default=None, choices=["on", "off"],
                    help="the extra blocks (stock proposal, configs[1] hipGraph, kernel legs, parity); default: on for N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--tunableop", default="off", choices=["on", "off"],
                    help="PyTorch TunableOp for the user callables' matmuls (tunes each GEMM shape once, in warm-up). "
                         "Off by default since round 3: the per-timestep matmuls of a model written `x @ W.t() + c` are "
                         "recorded on the lazy latents and evaluated inside the fused kernels, and tuning the one "
                         "[B*K, d] x [d, d] product left at time 0 took 200 s of the default run")
    ap.add_argument("--tunableop-file", default=None,
                    help="keep TunableOp's picks in this CSV: a later run that finds it skips the tuning trials "
                         "(used to profile without them)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and take the sharded code path even with one rank (test hook)")
    ap.add_argument("--mode", default=None, choices=["graph", "eager"],
                    help="graph: replay the whole ELBO as one hipGraph (aesmc_amd.graphs); eager: Python loop")
    ap.add_argument("--dry-run", action="store_true",
                    help="print what would be launched (N > 1: the torch.distributed.run command and the "
                         "environment it adds) as JSON and exit without touching the GPU")
    ap.add_argument("--grad", default=None, choices=["on", "off"],
                    help="record the autograd graph during the forward step (default: on where it fits in HBM)")
    return ap.parse_args(argv)


# ---- launcher --------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(argv, gpus, port=None, python=None):
    """The command `python bench.py --gpus N ...` runs for N > 1: one rank per GPU under
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the hostname may not resolve)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.join(ROOT, "bench.py")] + list(argv)


def child_environment(base=None):
    env = dict(os.environ if base is None else base)
    env.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")          # torchrun's default of 1 starves the CPU-side numpy draws
    env["AESMC_BENCH_CHILD"] = "1"
    return env


def resolve_scaling(args, world):
    """The curve `--gpus N` headlines: BASELINE.json's north star splits c4's B=1024 rows over the GPUs (strong);
    the other workloads keep their B per GPU.  With one GPU the two coincide."""
    if args.scaling is not None:
        return args.scaling
    return "strong" if (world > 1 and args.workload == "c4") else "weak"


def needs_launcher(args, environ=None):
    environ = os.environ if environ is None else environ
    return args.gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def self_launch(args, argv):
    """Nothing in this process has touched the GPU (importing torch does not): the N ranks are fresh
    children; their rank 0 prints the one JSON line on the stdout they inherit."""
    command = child_command(argv, args.gpus)
    if args.dry_run:
        added = {k: v for k, v in child_environment({}).items()}
        print(json.dumps({"launch": command, "environment_defaults": added, "scaling": resolve_scaling(args, args.gpus),
                          "workload": args.workload}))
        return 0
    print("bench.py: launching {} ranks: {}".format(args.gpus, " ".join(command)), file=sys.stderr, flush=True)
    return subprocess.run(command, env=child_environment()).returncode


# ---- models / CPU baseline -------------------------------------------------------------------------
def build_model(kind, dim, device, state, proposal="stock", callables="matmul", **model_kwargs):
    from aesmc_amd.testing import models
    cls = {"lgssm": models.LgssmNd, "nonlinear": models.NonlinearSsm, "gaussian": models.GaussianIwae,
           "learned_scale": models.LearnedScaleSsm, "lgssm1d_reference": models.ReferenceLgssm1d}[kind]
    if kind == "lgssm1d_reference":      # nothing passed down: the classes' own defaults (validate_args included)
        return cls(state=state).to(device)
    if kind == "gaussian":
        return cls(state=state, validate_args=False).to(device)
    if kind == "lgssm":
        model_kwargs = dict(model_kwargs, affine=(callables == "affine"))
    if kind == "nonlinear":     # d x d maps through K8 (the proposal net stays PyTorch's)
        model_kwargs = dict(model_kwargs, fused=(callables == "affine"))
    # validate_args=False: no per-call host sync inside torch.distributions (standard practice)
    model = cls(dim, seed=0, state=state, validate_args=False, **model_kwargs).to(device)
    if proposal == "tuned" and hasattr(model, "tune_proposal"):
        model.tune_proposal()
    return model


def cpu_baseline(kind, dim, B, K, T, algorithm="aesmc", proposal="stock", model_kwargs=None, budget_s=60.0):
    """The CPU port of the reference (oracle/reference_port.py) on a BOUNDED sample of the same workload: same model,
    K, T and d; batch rows (and, only if one row is still too slow, timesteps) are cut until one evaluation takes about a
    twelfth of `budget_s`.  SURVEY.md 8(d) / BASELINE.md section 3: the thread count is swept AT THE TIMED SIZE ({8, 16, 32,
    64, all} of the host's cores, ascending, one evaluation each, stopped once a count is clearly slower — PyTorch's
    default of one thread per core is far slower on big hosts for these small ops), then the MEDIAN of 5 evaluations at
    the best count; the spread and the sweep are printed with it."""
    import numpy as np
    import torch
    from oracle import reference_port
    started = time.perf_counter()
    cores = os.cpu_count() or 1
    candidates = sorted({t for t in (8, 16, 32, 64, cores) if t <= cores}) or [cores]
    model = build_model(kind, dim, torch.device("cpu"), reference_port, proposal, **(model_kwargs or {}))
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

    # ---- size the sample: a small piece at the workload's own T (the O(T^2) history re-gather decides the rate), timed
    # on the first candidate's threads, scaled to the evaluation's share of the budget
    per_run = budget_s / 12.0
    torch.set_num_threads(candidates[0])
    cal_t = T
    while cost(1, cal_t) > 3e9 and cal_t > 6:
        cal_t = max(6, cal_t // 2)
    cal_b = int(max(1, min(B, 4)))
    run(cal_b, cal_t)                       # warm-up (thread pool, allocator)
    cal_dt, _ = run(cal_b, cal_t)
    rate = cost(cal_b, cal_t) / max(cal_dt, 1e-6)
    b, t = B, T
    while cost(b, t) / rate > per_run and b > 1:
        b = max(1, b // 2)
    while cost(b, t) / rate > per_run and t > 10:
        t -= 5
    run(b, t)                               # warm-up at the sample's own size (the first pass grows the allocator)
    # the calibration piece is pessimistic (a handful of rows run less efficiently than dozens): grow the sample, in rows,
    # to what a WARM evaluation's measured time says fits its share
    dt, _ = run(b, t)
    while 2.3 * dt <= per_run and 2 * b <= B:
        b *= 2
        run(b, t)
        dt, _ = run(b, t)
    # ---- the thread sweep at the timed size
    sweep_seconds = {}
    for threads in candidates:
        torch.set_num_threads(threads)
        if threads != candidates[0]:
            run(min(b, 2), min(t, 10))      # (a new pool's first pass)
        sweep_seconds[threads], _ = run(b, t)
        best = min(sweep_seconds.values())
        # past the best count more threads only lose (at 256 threads these small ops ran 500 times slower than at 16)
        if sweep_seconds[threads] > best / 0.6 or time.perf_counter() - started > 0.5 * budget_s:
            break
    sweep = {k: b * K * t / max(v, 1e-6) for k, v in sweep_seconds.items()}
    # the FEWEST threads within 5 % of the best rate (near-ties go to the count that depends least on the host's mood)
    threads = min(k for k in sweep if sweep[k] >= 0.95 * max(sweep.values()))
    torch.set_num_threads(threads)
    run(min(b, 2), min(t, 10))
    runs = 5
    remaining = 1.3 * budget_s - (time.perf_counter() - started)
    if sweep_seconds[threads] * runs > remaining:      # a slow host: what fits, at least 3
        runs = max(3, int(remaining / max(sweep_seconds[threads], 1e-6)))
    timed = sorted(run(b, t) for _ in range(runs))
    dt, loss = timed[len(timed) // 2]
    return {"value": b * K * t / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": "median of {} forward ELBOs (after warm-up), B={} K={} T={} d={} ({} proposal), {:.1f} s each on {} "
                      "of {} host cores; oracle/reference_port.py (PyTorch-CPU + NumPy, keeps the reference's O(T^2) "
                      "history re-gather and per-row np.digitize loop)".format(runs, b, K, t, dim, proposal, dt, threads,
                                                                               cores),
            "spread": [b * K * t / timed[-1][0], b * K * t / timed[0][0]],
            "thread_sweep": {"particle_steps_per_sec_by_threads": {str(k): round(v, 1) for k, v in sweep.items()},
                             "piece": "B={} T={} (the timed sample itself)".format(b, t)},
            "host_cores": cores, "seconds": round(time.perf_counter() - started, 1), "loss": loss}


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


def _round(stats):
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()}


# ---- one workload ------------------------------------------------------------------------------------
class Context:
    """What every leg of the bench shares: the device, the process group and the kernel provider."""

    def __init__(self, args, device, rank, world, use_dist):
        self.args, self.device, self.rank, self.world, self.use_dist = args, device, rank, world, use_dist

    def barrier(self):
        import torch
        if self.use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, n):
        """n calls of fn between barrier + synchronize on both sides; max over ranks."""
        import torch
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = fn()
        self.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.device)
        self.last_per_rank_seconds = [float(dt.item())]
        if self.use_dist:
            import torch.distributed as dist
            every = [torch.zeros_like(dt) for _ in range(dist.get_world_size())]
            dist.all_gather(every, dt)          # each rank's own clock: a slow rank shows up by name in the JSON line
            self.last_per_rank_seconds = [float(t.item()) for t in every]
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), float(loss.detach())


def traffic_record(workload, proposal, kernel):
    """HBM bytes per launch of the roofline kernel from the PMC passes committed under profiles/
    (tools/pmc_workload.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per run, over the
    first timesteps of this same seeded workload; FETCH_SIZE doubled as calibrated on an
    identity-index launch).  Counters cannot be read from inside an unprofiled run, so the line
    carries the committed figure and says where it came from."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        table = json.load(fh)
    entry = table.get("{}:{}".format(workload, proposal), {}).get(kernel)
    if not entry:
        return None, None
    return entry.get("hbm_bytes_per_launch"), entry.get("source", "profiles/pmc_traffic.json")


def run_workload(ctx, name, proposal, steps, warmup, scaling="weak", want_backward=True, want_kernels=True,
                 mode=None, grad=None, callables=None):
    """Times `steps` forward ELBOs of one workload (after `warmup`), then forward+backward, then the
    per-kernel pass.  Returns a dict with the contract's numbers for this workload."""
    import numpy as np
    import torch
    import aesmc_amd
    from aesmc_amd import _kernels, distributed

    args, device, rank, world = ctx.args, ctx.device, ctx.rank, ctx.world
    description, kind, dim, B, K, T, model_kwargs = WORKLOADS[name]
    algorithm = ALGORITHM.get(name, "aesmc")
    if kind != "lgssm":
        proposal = "stock"
    callables = (callables or args.callables) if kind in ("lgssm", "nonlinear") else "matmul"
    if scaling == "strong":
        global_B = B
        lo, hi = distributed.shard_bounds(global_B, rank, world)
        local_B = hi - lo
    else:
        global_B, local_B = B * world, B
    provider = _kernels.get()
    torch.cuda.reset_peak_memory_stats(device)
    model = build_model(kind, dim, device, aesmc_amd.state, proposal, callables, **model_kwargs)
    observations = model.simulate(T, global_B, seed=1)          # same data on every rank ...
    observations = distributed.shard_observations(observations, rank, world)  # ... own rows only
    parts = (model.initial, model.transition, model.emission, model.proposal)
    np.random.seed(0)             # the same on every rank: the resampler draws the global uniform block and keeps its rows
    torch.manual_seed(rank)       # different per rank: independent proposal noise (rank 0: the single-GPU run's stream)

    if mode is None:      # small per-step sizes are host-bound in the eager loop
        mode = "graph" if (local_B * K <= GRAPH_PARTICLES and name not in NO_GRAD) else "eager"
    if grad is None:
        grad = "off" if name in NO_GRAD else "on"
    grad_mode = torch.enable_grad if grad == "on" else torch.no_grad
    if grad == "off":
        want_backward = False

    def step(backward=False):
        with grad_mode():
            if ctx.use_dist:
                loss = distributed.sharded_get_loss(observations, K, algorithm, *parts, global_batch_size=global_B,
                                                    rank=rank, world_size=world)
            else:
                loss = aesmc_amd.losses.get_loss(observations, K, algorithm, *parts)
            if backward:
                model.zero_grad(set_to_none=True)
                loss.backward()
                if ctx.use_dist:
                    distributed.all_reduce_gradients(list(model.parameters()))
            return loss.detach()

    # ---- the step: one hipGraph replay of the whole ELBO or the eager Python loop -----------------
    shard = (global_B, rank, world) if ctx.use_dist else None
    ran_as, graph_error, graphed = "eager", None, None
    if mode == "graph":
        try:
            from aesmc_amd import graphs
            with grad_mode():
                # check_flags=False: replays run back to back; the device status word (NaN weights,
                # degenerate rows, ...) is read once after each timed region instead of every step
                graphed = graphs.GraphedLoss(observations, K, algorithm, *parts, shard=shard, check_flags=False)
            ran_as = "hipgraph"
        except Exception as error:  # capture is an optimisation: report and fall back to eager
            graph_error = "{}: {}".format(type(error).__name__, error)
            torch.cuda.synchronize()
    forward = graphed if graphed is not None else step

    for _ in range(warmup):
        forward()
    seconds, loss = ctx.timed(forward, steps)
    per_rank_ms = [round(1e3 * t / steps, 4) for t in ctx.last_per_rank_seconds]
    if graphed is not None:
        graphed.check()
    out = {
        "workload": "{}: {}".format(name, description), "proposal": proposal if kind == "lgssm" else None,
        "callables": callables,
        "value": global_B * K * T * steps / seconds, "ms_per_step": 1e3 * seconds / steps, "loss": loss,
        "mode": ran_as, "graph_error": graph_error, "grad": grad, "scaling": scaling,
        "batch_per_gpu": local_B, "global_batch": global_B, "num_particles": K, "num_timesteps": T,
        "state_dim": dim, "algorithm": algorithm, "per_rank_ms_per_step": per_rank_ms,
        "peak_memory_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
    }

    if graphed is not None:  # the eager loop beside it, for the record
        for _ in range(2):      # (its first passes beside the capture's pool grow the allocator: not what a loop costs)
            step()
        n = max(2, steps // 2)
        se, _ = ctx.timed(step, n)
        out["eager_particle_steps_per_sec"] = global_B * K * T * n / se

    if want_backward:
        n = max(1, steps // 2)
        train_step = graphed_train = None
        if graphed is not None:
            try:
                from aesmc_amd import graphs
                graphed_train = graphs.GraphedLoss(observations, K, algorithm, *parts, backward=True, shard=shard,
                                                   check_flags=False)
                params = list(model.parameters())

                def train_step():
                    result = graphed_train()
                    if ctx.use_dist:
                        distributed.all_reduce_gradients(params)
                    return result
            except Exception as error:
                out["graph_error"] = "backward capture: {}: {}".format(type(error).__name__, error)
                torch.cuda.synchronize()
                train_step = graphed_train = None
        if train_step is None:
            def train_step():
                return step(backward=True)
        try:
            train_step()
            sb, _ = ctx.timed(train_step, n)
            if graphed_train is not None:
                graphed_train.check()
            out["fwd_bwd_particle_steps_per_sec"] = global_B * K * T * n / sb
            out["fwd_bwd_ms_per_step"] = 1e3 * sb / n
        except torch.OutOfMemoryError as error:
            out["fwd_bwd_error"] = "OutOfMemoryError: {}".format(str(error)[:200])
            model.zero_grad(set_to_none=True)
        del graphed_train, train_step
    out["peak_memory_GB"] = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)

    # ---- per-kernel timing: the same steps again with every launch noted, a sample of them replayed
    # back to back between two HIP events on the stream they are launched on ------------------------
    if want_kernels:
        provider.timer = _kernels.KernelTimer()
        try:
            for _ in range(max(1, min(steps, 3))):
                step()
            kernels = provider.timer.summary()
        finally:
            provider.timer = None
        out["kernels"] = {k: _round(v) for k, v in kernels.items()}
        if want_backward and "fwd_bwd_error" not in out:
            # the backward's launches, timed the same way (one eager forward + backward with every launch noted): K14 —
            # a linear-Gaussian timestep's whole backward — is 43 % of a training step's device time at c4
            provider.timer = _kernels.KernelTimer(keep=12)
            try:
                step(backward=True)
                backward = {k: v for k, v in provider.timer.summary().items() if k not in kernels}
            finally:
                provider.timer = None
            model.zero_grad(set_to_none=True)
            out["backward_kernels"] = {k: dict(_round(v), frac=round(v["GBps"] / HBM_PEAK_GBPS, 4))
                                       for k, v in backward.items()}
        # the roofline kernel: the resample gather (K3, or the fused step that contains it); a workload
        # that never resamples (c3, IWAE) is what BASELINE.json uses to isolate the fused log-weight
        # + log-sum-exp kernel (K1)
        wide = kernels.get("affine_normal_propagate_wide")
        if wide is not None and wide["avg_us"] * wide["launches"] >= 0.5 * sum(
                v["avg_us"] * v["launches"] for v in kernels.values()):
            key, label = "affine_normal_propagate_wide", None      # K17 + K18: priced against the matrix cores below
        elif "affine_normal_propagate_drawn" in kernels:
            key, label = "affine_normal_propagate_drawn", \
                "affine_propagate_item_kernel (K16: the resample gather, the proposal's noise and draw, the log-weight; " \
                "one work item per workgroup — affine_propagate_fused_kernel where weights' rows are strided)"
        elif "affine_normal_propagate_resampled" in kernels:
            key, label = "affine_normal_propagate_resampled", \
                "affine_logweight_kernel, DRAW + GATHER (the resample gather inside the propagation launch: K3 + K15)"
        elif "moved_GBps" in kernels.get("resample_step", {}):
            key, label = "resample_step", "ancestor_index_inv_kernel with payload (fused step: K2 + K3)"
        elif "resample_gather" in kernels:
            key, label = "resample_gather", "resample_gather_kernel (K3)"
        else:
            key, label = "logweight_lse", "logweight_lse_kernel (K1)"
        if key == "affine_normal_propagate_wide":
            out["roofline"] = wide_roofline_of(kernels[key], local_B, K, dim)
        else:
            out["roofline"] = roofline_of(kernels.get(key), label, name, proposal, key)
        if out["roofline"] is not None and use_smc_path(algorithm):
            # the WHOLE hot path against SURVEY.md 8(d)'s algorithmic figure (36 + 8 d B per particle-step: K1 16 +
            # K2 12 + K3 8 d + 8) over the timed ELBOs — what the fused launches are for (round 2: 0.24 of the peak)
            per_step = 36 + 8 * dim
            rate = per_step * B * K * T / (out["ms_per_step"] * 1e-3) / 1e9
            out["roofline"]["whole_path"] = {"algorithmic_bytes_per_particle_step": per_step, "achieved": round(rate, 1),
                                             "frac": round(rate / HBM_PEAK_GBPS, 4), "unit": "GB/s",
                                             "over": "ms_per_step of the timed region (this rank's rows)"}
        # the propagation kernels either side of it (K9 draws x_t, K10 weighs it), priced the same way
        others = [roofline_of(kernels.get(k), l, name, proposal, k) for k, l in (
            ("resample_step" if key != "resample_step" else "-", "ancestor_index_inv_kernel (K2: ancestor indices + row log-sum-exp)"),
            ("philox_normal_fill", "philox_normal_fill_kernel (the step's noise, torch's Philox stream)"),
            ("affine_normal_propagate", "affine_logweight_kernel, DRAW (K15: the draw and its log-weight)"),
            ("affine_normal_rsample", "affine_rsample_kernel (K9)"),
            ("affine_normal_logweight", "affine_logweight_kernel (K10)")) if k in kernels]
        if out["roofline"] is not None and others:
            out["roofline"]["path_kernels"] = [
                {f: o[f] for f in ("kernel", "avg_launch_us", "achieved", "frac", "algorithmic_bytes_per_launch",
                                   "traffic", "frac_traffic", "launches") if f in o} for o in others]
    del graphed, forward, step, model, observations
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def use_smc_path(algorithm):
    return algorithm == "aesmc"


def wide_roofline_of(stats, B, K, d):
    """The contract's `roofline` for a step on rows of 128 values (configs[4]): K17 + K18 — one C-ABI call, two launches —
    are three d x d maps per particle on the fp32 matrix cores, so the bound is "mfma": 3 * 2 B K d^2 FLOPs over the
    call's time against the dense fp32 matrix peak; the HBM figure of the same call rides beside it."""
    flops = 3 * 2.0 * B * K * d * d
    us = stats["avg_us"]
    return {"kernel": "affine_wide_draw_kernel + affine_wide_emission_kernel (K17 + K18: gather, both maps of x_{t-1}, "
                      "draw; emission map and log-weight — fp32 matrix cores)",
            "bound": "mfma", "achieved": round(flops / us / 1e6, 1), "peak": MFMA_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(min(flops / us / 1e6 / MFMA_FP32_PEAK_TFLOPS, 1.0), 4), "pricing": "algorithmic",
            "avg_launch_us": round(us, 2), "flops_per_launch": flops, "launches": stats["launches"],
            "algorithmic_bytes_per_launch": stats["bytes_per_launch"], "hbm_GBps": round(stats["GBps"], 1),
            "frac_hbm": round(stats["GBps"] / HBM_PEAK_GBPS, 4), "traffic": None}


def roofline_of(stats, label, workload, proposal, key):
    """The contract's `roofline` object for one kernel's timing record.  `achieved` prices SURVEY.md
    8(d)'s ALGORITHMIC bytes (a full read of the gather's source).  A collapsed particle system
    fetches only the surviving rows, so the bytes that had to move are priced beside it, and when
    fewer than 30 % of the ancestors survive `achieved` / `frac` are the moved-bytes figures
    (`pricing` says which): a gather that mostly re-reads one cached row is not HBM traffic."""
    if not stats:
        return None
    algorithmic = stats["GBps"]
    out = {"kernel": label, "bound": "hbm", "achieved": round(algorithmic, 1), "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": round(algorithmic / HBM_PEAK_GBPS, 4), "pricing": "algorithmic",
           "avg_launch_us": round(stats["avg_us"], 2),
           "algorithmic_bytes_per_launch": stats["bytes_per_launch"], "launches": stats["launches"],
           "frac_algorithmic": round(algorithmic / HBM_PEAK_GBPS, 4)}
    traffic, source = traffic_record(workload, proposal, key)
    out["traffic"] = traffic
    out["traffic_source"] = source
    if traffic:     # HBM bytes the counters saw per launch, over this run's launch time
        out["achieved_traffic"] = round(traffic / stats["avg_us"] / 1e3, 1)
        out["frac_traffic"] = round(min(traffic / stats["avg_us"] / 1e3 / HBM_PEAK_GBPS, 1.0), 4)
    if "unique_ancestor_fraction" in stats:
        out["unique_ancestor_fraction"] = round(stats["unique_ancestor_fraction"], 4)
        out["moved_bytes_per_launch"] = stats["moved_bytes_per_launch"]
        out["achieved_moved_bytes"] = round(stats["moved_GBps"], 1)
        out["frac_moved_bytes"] = round(stats["moved_GBps"] / HBM_PEAK_GBPS, 4)
        if "ess_over_k" in stats:
            out["ess_over_k"] = round(stats["ess_over_k"], 4)
        if stats["unique_ancestor_fraction"] < 0.3:
            out["achieved"], out["frac"], out["pricing"] = out["achieved_moved_bytes"], out["frac_moved_bytes"], \
                "moved bytes (collapsed particle system: fewer than 30 % of the ancestors survive)"
    out["frac"] = min(out["frac"], 1.0)
    return out


# ---- kernel legs: the shapes BASELINE.json uses to isolate single kernels ---------------------------
def kernel_legs(ctx):
    """K1 at configs[2]'s shape (combine + log-sum-exp over [4096, 8192]); K3 at configs[4]'s shape
    (d=128, K=16384) on indices of a healthy and of a collapsed particle system; the fused step at
    configs[1]'s and at the 8-GPU shard's shape.  Synthetic operands as SURVEY.md 8(d) prescribes for
    kernel isolation: log_weight = s * randn (s = 1: ESS/K ~ 0.37), value = randn."""
    import torch
    from aesmc_amd import _kernels
    k = _kernels.get()
    dev = ctx.device
    gen = torch.Generator(device=dev).manual_seed(0)

    def timeit(fn, reps=20, replays=5):
        """Device time per call: `reps` calls captured in one hipGraph, replayed between two HIP events
        (no host gaps: the small launches here run shorter than Python can issue them)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(replays):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / (reps * replays)

    def leg(us, nbytes, **more):
        gbps = nbytes / us / 1e3
        out = dict({"avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": nbytes,
                    "achieved": round(gbps, 1), "frac": round(min(gbps / HBM_PEAK_GBPS, 1.0), 4),
                    "frac_algorithmic": round(gbps / HBM_PEAK_GBPS, 4), "pricing": "algorithmic"}, **more)
        if out.get("unique_ancestor_fraction", 1.0) < 0.3:      # as roofline_of: a collapsed gather is priced by what moved
            out["achieved"], out["frac"] = out["achieved_moved_bytes"], out["frac_moved_bytes"]
            out["pricing"] = "moved bytes (collapsed indices)"
        return out

    legs = {}
    B, K = 4096, 8192
    a, b, c = [torch.randn(B, K, device=dev, generator=gen) for _ in range(3)]
    legs["K1_c3_combine_lse"] = leg(timeit(lambda: k.logweight_lse(a, b, c)), B * K * 16 + 4 * B,
                                    shape="B=4096 K=8192 (configs[2])")
    legs["K1_c3_lse_only"] = leg(timeit(lambda: k.logweight_lse(a, None, None, want_lw=False)), B * K * 4 + 4 * B,
                                 shape="B=4096 K=8192, row log-sum-exp of one input")
    del a, b, c
    for label, (B, K, d) in (("c5", (64, 16384, 128)), ("c4", (1024, 4096, 10))):
        x = torch.randn(B, K, d, device=dev, generator=gen)
        u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
        for s in (1.0, 5.0):
            lw = s * torch.randn(B, K, device=dev, generator=gen)
            idx = k.ancestor_index(lw, u)
            unique = (int((idx[:, 1:] != idx[:, :-1]).sum().item()) + B) / (B * K)
            us = timeit(lambda: k.gather(x, idx))
            moved = B * K * 8 + (1 + unique) * x.numel() * 4
            legs["K3_{}_s{:g}".format(label, s)] = leg(
                us, B * K * (8 + 8 * d), shape="B={} K={} d={}".format(B, K, d),
                unique_ancestor_fraction=round(unique, 4), achieved_moved_bytes=round(moved / us / 1e3, 1),
                frac_moved_bytes=round(min(moved / us / 1e3 / HBM_PEAK_GBPS, 1.0), 4))
        del x
    for label, (B, K, d) in (("c2", (256, 1024, 10)), ("c4s", (128, 4096, 10)), ("c4", (1024, 4096, 10))):
        x = torch.randn(B, K, d, device=dev, generator=gen)
        u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
        lw = torch.randn(B, K, device=dev, generator=gen)
        idx = k.ancestor_index(lw, u)
        unique = (int((idx[:, 1:] != idx[:, :-1]).sum().item()) + B) / (B * K)
        us = timeit(lambda: k.resample_step(lw, u, x, want_lse=True))
        moved = B * K * 12 + (1 + unique) * x.numel() * 4
        legs["step_{}_s1".format(label)] = leg(
            us, B * K * (20 + 8 * d) + 8 * B, shape="B={} K={} d={} (fused step: K2 + K3)".format(B, K, d),
            unique_ancestor_fraction=round(unique, 4), achieved_moved_bytes=round(moved / us / 1e3, 1),
            frac_moved_bytes=round(min(moved / us / 1e3 / HBM_PEAK_GBPS, 1.0), 4))
        del x
    # linear-Gaussian propagation (K9 draw, K10 log-weight, K12 its backward, K11 the draw's adjoint) on
    # N(0,1) particles: algorithmic bytes = the [B,K,d] tensors each must read and write once
    for label, (B, K, d) in (("c4", (1024, 4096, 10)), ("c2", (256, 1024, 10))):
        make = lambda *shape: torch.randn(*shape, device=dev, generator=gen)
        x_prev, x, eps, y, off = make(B, K, d), make(B, K, d), make(B, K, d), make(B, d), make(B, d)
        eye = torch.eye(d, device=dev)
        A, C, Q = 0.9 * eye + 0.01 * make(d, d), eye + 0.01 * make(d, d), 0.45 * eye + 0.01 * make(d, d)
        scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
        terms = ((A, None), (C, None), (Q, off))
        shape = "B={} K={} d={}".format(B, K, d)
        N = B * K
        legs["K9_{}_affine_rsample".format(label)] = leg(
            timeit(lambda: k.affine_rsample(x_prev, Q, off, eps, scales[2])), 4 * N * 3 * d, shape=shape)
        legs["K10_{}_affine_logweight".format(label)] = leg(
            timeit(lambda: k.affine_logweight(x_prev, x, y, *terms, scales)), 4 * N * (2 * d + 1), shape=shape)
        lw = k.affine_logweight(x_prev, x, y, *terms, scales)
        lse = k.logweight_lse(lw, None, None, want_lw=False)[1]
        grad_lse = torch.full_like(lse, -1.0 / B)
        need = [True, True, False, True, False, True, False, True, True, False, False, False]
        legs["K12_{}_affine_logweight_backward".format(label)] = leg(
            timeit(lambda: k.affine_logweight_backward(x_prev, x, y, *terms, scales, need, lw=lw, lse=lse,
                                                       grad_lse=grad_lse), reps=5),
            4 * N * (5 * d + 1), shape=shape + " (x_prev, x, lw in; both latent gradients and the proposal "
                                               "offset's per-particle gradient out)")
        legs["K11_{}_affine_adjoint".format(label)] = leg(
            timeit(lambda: k.particle_affine_backward(eps, x_prev, Q), reps=5), 4 * N * 3 * d, shape=shape)
        del x_prev, x, eps, lw
    # the FIRST timestep in one launch (K20: the BATCH_EXPANDED proposal's transposed draw, the emission's location, the
    # log-weight) beside the three launches it stands for (K6 + K8 + K5), the headline model's parameter shapes
    for label, (B, K, d) in (("c4", (1024, 4096, 10)), ("c2", (256, 1024, 10))):
        make = lambda *shape: torch.randn(*shape, device=dev, generator=gen)
        full = lambda t: (t if t.dim() < 2 else t.unsqueeze(1)).expand(B, K, d)
        eps, out_x, C = make(K, B, d), torch.empty(B, K, d, device=dev), torch.eye(d, device=dev) + 0.01 * make(d, d)
        loc_q, scale_q = full(make(B, d)), full(torch.tensor(0.7, device=dev))
        loc_p, scale_p = full(torch.zeros(d, device=dev)), full(torch.ones(d, device=dev))
        obs, scale_g = full(make(B, d)), full(torch.tensor(0.5, device=dev))

        def three():
            x = k.normal_rsample(eps.transpose(0, 1), loc_q, scale_q)
            return k.normal_logweight(x, loc_p, scale_p, obs, k.particle_affine(x, C, None), scale_g, loc_q, scale_q)

        one = lambda: k.affine_initial_step(eps, loc_q, scale_q, loc_p, scale_p, obs, C, None, scale_g, out_x)
        if one() is not None:
            legs["K20_{}_first_step".format(label)] = leg(
                timeit(one, reps=5), 4 * B * K * (2 * d + 1), shape="B={} K={} d={}".format(B, K, d),
                three_launches_us=round(timeit(three, reps=5), 2))
        del eps, out_x
    # configs[4]'s extent: the step on the fp32 matrix cores (K17 + K18), priced against their dense peak
    B, K, d = 64, 16384, 128
    make = lambda *shape: torch.randn(*shape, device=dev, generator=gen)
    x_prev, eps, y, off = make(B, K, d), make(B, K, d), make(B, d), make(B, d)
    eye = torch.eye(d, device=dev)
    A, C, Q = 0.9 * eye + 0.01 * make(d, d), 0.1 * make(d, d), 0.45 * eye + 0.01 * make(d, d)
    scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
    idx = k.ancestor_index(torch.randn(B, K, device=dev, generator=gen), torch.rand(B, device=dev, dtype=torch.float64,
                                                                                   generator=gen))
    out_x = torch.empty_like(x_prev)
    wide = lambda: k.affine_propagate_wide(x_prev, eps, y, (A, None), (C, None), (Q, off), scales, out_x, ancestors=idx)
    if wide() is not None:
        us = timeit(wide, reps=5)
        flops = 3 * 2.0 * B * K * d * d
        nbytes = 4 * B * K * (3 * d + 1) + 8 * B * K
        legs["K17_K18_c5_wide_step"] = {
            "avg_launch_us": round(us, 1), "shape": "B={} K={} d={} (gather + draw from given noise + log-weight, two launches)".format(B, K, d),
            "bound": "mfma", "flops_per_step": flops, "achieved": round(flops / us / 1e6, 1), "unit": "TFLOP/s",
            "peak": MFMA_FP32_PEAK_TFLOPS, "frac": round(flops / us / 1e6 / MFMA_FP32_PEAK_TFLOPS, 4),
            "algorithmic_bytes_per_step": nbytes, "hbm_GBps": round(nbytes / us / 1e3, 1)}
    del x_prev, eps, out_x
    torch.cuda.empty_cache()
    return legs


def parity_block(ctx):
    """float32 ancestor indices against the reference's own (fixtures captured from the imported
    reference by oracle/capture_golden.py).  The reference builds the CDF in float32 (NumPy / SciPy),
    this library in float64: float64 inputs agree exactly, float32 inputs
    wherever no comparison sits within float32 rounding noise of flipping — the rate is stated here."""
    import glob
    import numpy as np
    import torch
    from aesmc_amd import _kernels
    k = _kernels.get()
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resampler_*.npz")) +
                       glob.glob(os.path.join(ROOT, "tests", "golden", "lgssm10d_k1024_smc_f32.npz"))):
        data = np.load(path)
        name = os.path.basename(path)[:-4]
        if "log_weight" in data.files:
            cases = [(data["log_weight"], data["uniform"], data["out_idx"])]
        else:
            steps = len([f for f in data.files if f.startswith("out_idx_")])
            cases = [(data["out_log_weights_{}".format(t)], data["uniform_{}".format(t)], data["out_idx_{}".format(t)])
                     for t in range(steps)]
        total = wrong = worst = 0
        for lw, u, want in cases:
            if not np.isfinite(lw).all() or lw.dtype not in (np.float32, np.float64):
                continue
            got = k.ancestor_index(torch.from_numpy(lw).to(ctx.device),
                                   torch.from_numpy(np.asarray(u, dtype=np.float64).reshape(-1)).to(ctx.device))
            delta = (got.cpu().numpy() - want.astype(np.int64))
            total += delta.size
            wrong += int((delta != 0).sum())
            worst = max(worst, int(np.abs(delta).max()))
        if total:
            out[name] = {"dtype": str(cases[0][0].dtype), "num_particles": int(cases[0][0].shape[1]),
                         "indices": total, "mismatches": wrong, "max_abs_delta": worst,
                         "flip_rate": wrong / total}
    k.read_flags(ctx.device)
    return out


def fixture_parity_block(ctx):
    """Every float32 SMC run captured from the reference (tests/golden/), replayed on the device draw for draw
    with the bench's own model statement (AffineNormal callables): what the device ACHIEVES, free-running and
    teacher-forced (aesmc_amd/testing/parity.py) — per-step index agreement, per-step log-weight error with the
    reference's ancestors substituted, |delta log Z|."""
    from aesmc_amd import state
    from aesmc_amd.testing import parity
    from tests.golden_io import Golden
    out = {}
    for name in ("lgssm10d_k1024_smc_f32", "lgssm10d_k4096_smc_f32", "lgssm3d_smc_f32", "c1_lgssm1d_smc_f32"):
        try:
            case = Golden(name)
        except FileNotFoundError:
            continue
        affine = case.meta["model"] == "lgssm_nd"
        parts, _ = case.build_parts(state, ctx.device, affine=affine)
        steps = case.meta["num_timesteps"] - 1
        reference = {"log_weights": case.series("out_log_weights"), "indices": case.series("out_idx")[:steps],
                     "lml": case["out_lml"]}
        got = parity.float32_fixture_parity(parts, case.observations(ctx.device), case.meta["num_particles"],
                                            case.tape(), reference)
        got = {k: ([round(x, 6) if isinstance(x, float) else x for x in v] if isinstance(v, list) else v)
               for k, v in got.items()}
        got["num_particles"], got["num_timesteps"], got["callables"] = case.meta["num_particles"], steps + 1, \
            "affine" if affine else "reference style"
        out[name] = got
    return out


def summary_of(out, head, extras):
    """Every claimed number of the run in well under 1.5 KB, as the last key of the JSON line: [particle-steps/s, ms per
    ELBO] per workload (replayed; `eager` beside it), forward + backward rates, per-kernel roofline fractions, the
    strong- and weak-scaling projections from one GPU, the float32 index flip rates against the reference's fixtures."""
    def rate(x):
        return None if x is None else float("%.4g" % x)

    def pair(leg):
        return None if not leg or leg.get("value") is None else [rate(leg["value"]), round(leg["ms_per_step"], 3)]

    summary = {"head": pair(head), "head_eager": rate(head.get("eager_particle_steps_per_sec")),
               "head_fwd_bwd": rate(head.get("fwd_bwd_particle_steps_per_sec"))}
    fracs = {}
    roof = head.get("roofline") or {}
    if roof:
        fracs["head_kernel"] = roof.get("frac")
        if roof.get("whole_path"):
            fracs["whole_path"] = roof["whole_path"]["frac"]
        for other in roof.get("path_kernels", []):
            if other["kernel"].startswith("ancestor_index"):
                fracs["K2"] = other["frac"]
                summary["K2_us"] = other["avg_launch_us"]
    for name, stats in (head.get("backward_kernels") or {}).items():
        if name.startswith("affine_step_backward"):
            fracs["K14"] = stats["frac"]
            summary["K14_us"] = round(stats["avg_us"], 1)
    legs = extras.get("kernel_legs") or {}
    for label, key in (("K1_c3", "K1_c3_combine_lse"), ("K3_c4", "K3_c4_s1"), ("K3_c5", "K3_c5_s1"),
                       ("K17_K18_mfma", "K17_K18_c5_wide_step")):
        if key in legs:
            fracs[label] = legs[key]["frac"]
    if "K20_c4_first_step" in legs:      # [the first timestep in one launch, the three launches it stands for] in us
        summary["K20_us"] = [legs["K20_c4_first_step"]["avg_launch_us"], legs["K20_c4_first_step"]["three_launches_us"]]
    summary["frac"] = fracs
    for label, key in (("c2", "c2_hipgraph"), ("c2ref", "reference_models"), ("c4_matmul", "matmul_callables"),
                       ("c4nl", "c4nl"), ("c5", "c5")):
        leg = extras.get(key)
        if leg:
            summary[label] = pair(leg) + [rate(leg.get("eager_particle_steps_per_sec")),
                                          rate(leg.get("fwd_bwd_particle_steps_per_sec"))]
    if summary.get("c2") or summary.get("c5"):
        summary["legend"] = "per workload: [replayed particle-steps/s, ms per ELBO, eager loop, fwd+bwd]"
    if (extras.get("c5") or {}).get("roofline"):
        fracs["c5_mfma"] = extras["c5"]["roofline"].get("frac")
    projection = extras.get("strong_scaling_projection")
    if projection:
        # strong: B = 1024 split over N GPUs (efficiency = t(1024) / (N t(1024 / N)), one GPU running one shard);
        # weak: every GPU keeps B = 1024 — the batch shard has no data-path collective but one 4-byte all-reduce per ELBO,
        # so its projection is t / (t + all-reduce)
        summary["strong_eff"] = {n: e["projected_efficiency"] for n, e in projection.items()}
        allreduce_ms = 1e-3 * (out.get("allreduce_us_per_elbo") or 25.0)
        summary["weak_eff"] = round(head["ms_per_step"] / (head["ms_per_step"] + allreduce_ms), 4)
    flips = extras.get("index_parity_vs_reference_fixtures") or {}
    if flips:
        summary["fp32_flip_rate"] = {name.replace("resampler_", ""): float("%.3g" % entry["flip_rate"])
                                     for name, entry in flips.items() if entry["dtype"] == "float32"}
        summary["fp64_flips"] = sum(entry["mismatches"] for entry in flips.values() if entry["dtype"] == "float64")
    if out.get("cpu_baseline"):
        summary["cpu"] = rate(out["cpu_baseline"]["value"])
    return summary


# ---- main ----------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if needs_launcher(args):
        sys.exit(self_launch(args, argv))
    if args.dry_run:
        print(json.dumps({"launch": None, "rank": int(os.environ.get("RANK", "0")),
                          "world_size": int(os.environ.get("WORLD_SIZE", "1")), "workload": args.workload,
                          "scaling": resolve_scaling(args, int(os.environ.get("WORLD_SIZE", "1")))}))
        return

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import aesmc_amd  # noqa: F401
    from aesmc_amd import _kernels

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    args.scaling = resolve_scaling(args, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    rccl_world, allreduce_us = None, None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)          # creates the communicator now (and prints its banner)
            torch.cuda.synchronize()
        rccl_world = dist.get_world_size()
        # what the data path's one collective costs: the all-reduce of sum_b log Z_b, once per ELBO
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        a.record()
        for _ in range(50):
            dist.all_reduce(warm)
        b.record()
        torch.cuda.synchronize()
        allreduce_us = round(a.elapsed_time(b) * 1e3 / 50, 2)
    assert _kernels.get().name == "hip"
    ctx = Context(args, device, rank, world, use_dist)

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

    extras_on = (args.extras or ("on" if world == 1 else "off")) == "on"
    seconds = {}          # wall time of every section of this run (the default run must stay within minutes)
    clock = [time.perf_counter()]

    def lap(label):
        now = time.perf_counter()
        seconds[label] = round(now - clock[0], 1)
        clock[0] = now
    head = run_workload(ctx, args.workload, args.proposal, args.steps, args.warmup, scaling=args.scaling,
                        want_backward=not args.no_backward, mode=args.mode, grad=args.grad)
    description, kind, dim, B, K, T, model_kwargs = WORKLOADS[args.workload]
    lap("headline workload")

    out = {
        "metric": "particle_steps_per_sec", "value": head["value"], "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": head["workload"], "proposal": head["proposal"],
                   "callables": {"affine": "AffineNormal(source, weight, scale, offset), the proposal's with defer_draw=True: "
                                           "resampling gather, draw (noise from torch's Philox stream, inside the launch) "
                                           "and log-weight in one kernel (K16; below 0.5M particles K15 through the "
                                           "ancestors behind a noise launch), a step's backward in one (K14, through the "
                                           "ancestors) (aesmc_amd/linear_gaussian.py)",
                                 "matmul": "Normal(source @ weight.T + offset, scale): PyTorch matmuls, then K6 / K5"
                                 }[head["callables"]],
                   "batch_per_gpu": head["batch_per_gpu"], "global_batch": head["global_batch"],
                   "num_particles": K, "num_timesteps": T, "state_dim": dim,
                   "parallelism": "batch-shard x{} (one RCCL all-reduce of sum log Z per ELBO)".format(world),
                   "pytorch": "TunableOp {} for the user callables' matmuls; distributions built with "
                              "validate_args=False".format(args.tunableop),
                   "hip_runtime": "DEBUG_CLR_GRAPH_PACKET_CAPTURE={} (0: captured memset nodes keep stream "
                                  "order on ROCm 7.0)".format(os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")),
                   "step": "one forward ELBO, get_loss(..., '{}'), ".format(head["algorithm"]) +
                           ("autograd graph recorded" if head["grad"] == "on" else "torch.no_grad()") +
                           (", all T timesteps replayed as one hipGraph" if head["mode"] == "hipgraph"
                            else ", eager Python loop")},
        "rccl_world_size": rccl_world, "allreduce_us_per_elbo": allreduce_us,
        "per_rank_ms_per_step": head["per_rank_ms_per_step"],      # each rank's own clock over the timed region (value uses the max)
        "loss": head["loss"], "mode": head["mode"], "graph_error": head["graph_error"],
        "tunableop": args.tunableop, "peak_memory_GB": head["peak_memory_GB"],
        "eager_particle_steps_per_sec": head.get("eager_particle_steps_per_sec"),
        # the same model written as the reference writes it — Normal(x @ W.t() + c, s), no library class — replayed and in
        # the plain eager loop (filled in below from the `matmul_callables` leg when the extras run)
        "reference_style_particle_steps_per_sec": None,
        "reference_style_eager_particle_steps_per_sec": None,
        "fwd_bwd_particle_steps_per_sec": head.get("fwd_bwd_particle_steps_per_sec"),
        "fwd_bwd_error": head.get("fwd_bwd_error"),
        "roofline": head.get("roofline"),
        "kernels": head.get("kernels"),
        "backward_kernels": head.get("backward_kernels"),
    }

    if head["callables"] == "matmul":
        out["reference_style_particle_steps_per_sec"] = head["value"]
        out["reference_style_eager_particle_steps_per_sec"] = head.get("eager_particle_steps_per_sec") \
            if head["mode"] == "hipgraph" else head["value"]

    extras = {}
    if world > 1 and (args.extras or "on") == "on":
        # the other curve beside the headline: weak scaling (every rank owns the workload's B rows) under a strong
        # headline, the split of the workload's B rows under a weak one
        other = "weak" if args.scaling == "strong" else "strong"
        beside = run_workload(ctx, args.workload, args.proposal, args.steps, args.warmup, scaling=other,
                              want_backward=False, want_kernels=False)
        extras[other + "_scaling"] = {key: beside[key] for key in
                                      ("value", "ms_per_step", "batch_per_gpu", "global_batch", "mode", "loss")}
    if extras_on and world == 1:
        def brief(result):
            keep = ("workload", "proposal", "callables", "value", "ms_per_step", "loss", "mode", "grad", "graph_error",
                    "eager_particle_steps_per_sec", "fwd_bwd_particle_steps_per_sec", "fwd_bwd_error",
                    "peak_memory_GB", "roofline")
            return {key: result.get(key) for key in keep}
        if kind == "lgssm":
            other = "stock" if args.proposal == "tuned" else "tuned"
            extras["{}_proposal".format(other)] = brief(run_workload(
                ctx, args.workload, other, max(2, args.steps // 2), min(args.warmup, 2), want_backward=False,
                want_kernels=False, mode=args.mode, grad=args.grad))
            lap("other proposal")
        if kind == "lgssm" and args.callables == "affine" and dim <= 16:
            # the same workload with the locations materialised by PyTorch matmuls (what rounds 1 and 2 timed)
            extras["matmul_callables"] = brief(run_workload(
                ctx, args.workload, args.proposal, 3, 1,
                want_backward=not args.no_backward, want_kernels=False, mode=args.mode, grad=args.grad,
                callables="matmul"))
            leg = extras["matmul_callables"]
            out["reference_style_particle_steps_per_sec"] = leg["value"]
            out["reference_style_eager_particle_steps_per_sec"] = leg.get("eager_particle_steps_per_sec") \
                if leg["mode"] == "hipgraph" else leg["value"]
            lap("matmul callables")
        if args.workload != "c2":
            extras["c2_hipgraph"] = brief(run_workload(ctx, "c2", args.proposal, 20, 5,
                                                       want_backward=not args.no_backward))
            lap("c2")
        if args.workload != "c2ref":
            # the reference's own model classes, unedited (Python-number scales, default validate_args): replayed as a
            # hipGraph and in the eager loop (`eager_particle_steps_per_sec`) — the day-one experience of a user who switches
            extras["reference_models"] = brief(run_workload(ctx, "c2ref", "stock", 20, 5, want_kernels=False,
                                                            want_backward=not args.no_backward))
            lap("reference-style models")
        if args.workload == "c4":
            # BASELINE.json configs[4]: rows of 128 values, forward only (autograd retention exceeds HBM); its
            # roofline is the matrix-core step K17 + K18
            extras["c5"] = brief(run_workload(ctx, "c5", args.proposal, 2, 1, want_backward=False))
            lap("c5")
        if args.workload == "c4":
            # BASELINE.json configs[3]'s model (nonlinear SSM, learned proposal net) on one GPU's shard of it, forward
            # AND backward: the d x d maps through K8 / K11, the proposal net through K13 / K13b (round 6: until then the
            # net was PyTorch's cat + Linear + tanh + Linear — 173 us of a 260 us timestep — and its two weight-gradient
            # GEMMs, a 524 288-long contraction onto a few workgroups, 898 + 822 us of the backward's 2.1 ms per timestep
            # under hipBLASLt's default picks: this leg then ran under TunableOp, ~50 s of tuning; no longer needed)
            extras["c4nl"] = brief(run_workload(ctx, "c4nl", "stock", 3, 2, want_backward=not args.no_backward,
                                                want_kernels=False, callables="affine"))
            extras["c4nl"]["tunableop"] = "off (the proposal net runs in the library's own kernels K13 / K13b)"
            # What one GPU's shard of the north-star batch costs on THIS device (global B = 1024 split over N GPUs:
            # B / N rows here): the strong-scaling curve, the one all-reduce of sum log Z per ELBO aside.
            # projected_efficiency = t(B = 1024) / (N * t(B / N)).
            projection = {}
            for shard_name, n in (("c4x2", 2), ("c4x4", 4), ("c4s", 8)):
                shard = run_workload(ctx, shard_name, args.proposal, 5, 2, want_backward=False, want_kernels=False)
                entry = {"batch_per_gpu": shard["batch_per_gpu"], "ms_per_elbo": round(shard["ms_per_step"], 3),
                         "mode": shard["mode"], "particle_steps_per_sec_per_gpu": shard["value"]}
                entry["projected_efficiency"] = round(head["ms_per_step"] / (n * entry["ms_per_elbo"]), 3)
                projection["N={}".format(n)] = entry
            extras["strong_scaling_projection"] = projection
            lap("c4nl + strong-scaling projection")
            if head["mode"] == "eager":
                # the same ELBO captured once and replayed (aesmc_amd.graphs.GraphedLoss): what is left of the host
                extras["c4_hipgraph"] = brief(run_workload(ctx, "c4", args.proposal, 5, 2, want_kernels=False, mode="graph",
                                                           want_backward=not args.no_backward))
                lap("c4 as a hipGraph")
        extras["kernel_legs"] = kernel_legs(ctx)
        lap("kernel legs")
        extras["index_parity_vs_reference_fixtures"] = parity_block(ctx)
        extras["fp32_fixture_parity"] = fixture_parity_block(ctx)
        lap("parity blocks")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, dim, B, K, T, head["algorithm"],
                                           args.proposal if kind == "lgssm" else "stock", model_kwargs)
        lap("cpu baseline")
    if extras:
        extras["bench_seconds"] = seconds
        out["extras"] = extras
    out["summary"] = summary_of(out, head, extras)      # LAST key: what survives a reader that keeps the line's tail
    if use_dist:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
