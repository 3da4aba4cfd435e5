"""K20 (the first timestep in one launch) against the three launches it stands for (K6 transposed draw, K8 emission
location, K5 log-weight) at a bench shape; hipGraph-timed, the bench model's parameter shapes (proposal location one row
per batch element, scalar proposal / emission scales, per-column prior).  python tools/initial_step_bench.py [B K d]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels  # noqa: E402

B, K, d = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 4096, 10))]
dev = torch.device("cuda", 0)
k = _kernels.get()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)
SETS = 4
eps = [r(K, B, d) for _ in range(SETS)]
out_x = [torch.empty(B, K, d, device=dev) for _ in range(SETS)]
y, loc_q_rows = r(B, d), r(B, d)
C = torch.eye(d, device=dev) + 0.01 * r(d, d)
full = lambda t: (t if t.dim() < 2 else t.unsqueeze(1)).expand(B, K, d)
loc_q, scale_q = full(loc_q_rows), full(torch.tensor(0.7, device=dev))
loc_p, scale_p = full(torch.zeros(d, device=dev)), full(torch.ones(d, device=dev))
obs, scale_g = full(y), full(torch.tensor(0.5, device=dev))


def three(i):
    x = k.normal_rsample(eps[i].transpose(0, 1), loc_q, scale_q)
    loc_g = k.particle_affine(x, C, None)
    return k.normal_logweight(x, loc_p, scale_p, obs, loc_g, scale_g, loc_q, scale_q)


def one(i):
    return k.affine_initial_step(eps[i], loc_q, scale_q, loc_p, scale_p, obs, C, None, scale_g, out_x[i])


def timeit(fn, replays=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(SETS):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for i in range(SETS):
            fn(i)
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (SETS * replays)


print("B={} K={} d={}".format(B, K, d))
for name, fn in (("K6 + K8 + K5", three), ("K20", one), ("K6 + K8 + K5", three), ("K20", one)):
    us = timeit(fn)
    nbytes = 4 * (2 * B * K * d + B * K)
    print("{:16s} {:8.1f} us   ({:.2f} TB/s of the one launch's {:.0f} MB)".format(name, us, nbytes / us / 1e6, nbytes / 1e6))
