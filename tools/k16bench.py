"""Times the forward step's launches at a bench shape: K2 alone, torch's normal_, K15, K15 through the ancestors
(K3 folded in), K16 (gather and noise inside).  hipGraph-timed, N(0,1) operands, healthy ancestries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops, _philox  # noqa: E402

B, K, d = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 4096, 10))]
dev = torch.device("cuda", 0)
k = _kernels.get()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)
SETS = 6      # distinct operand sets, so caches are as cold as in the workload
x_prev = [r(B, K, d) for _ in range(SETS)]
eps = [r(B, K, d) for _ in range(SETS)]
out_x = [torch.empty(B, K, d, device=dev) for _ in range(SETS)]
lw = [r(B, K) for _ in range(SETS)]
u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
y = r(B, d)
eye = torch.eye(d, device=dev)
A, C, Q = 0.9 * eye + 0.01 * r(d, d), eye + 0.01 * r(d, d), 0.45 * eye + 0.01 * r(d, d)
off_q = r(B, d)
terms = ((A, None), (C, None), (Q, off_q))
scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
idx = [_ops.ancestor_index(w, u) for w in lw]
res = _philox.reserve(B * K * d, dev)


def timeit(fn, replays=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(SETS):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for rep in range(3):
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
    return a.elapsed_time(b) * 1e3 / (3 * SETS * replays)


N = B * K
rows = [("K2 ancestor_index + lse (resample_step, no payload)", lambda i: k.resample_step(lw[i], u, None, True), N * 12),
        ("K2 + K3 fused step", lambda i: k.resample_step(lw[i], u, x_prev[i], True), N * (8 * d + 20)),
        ("torch normal_", lambda i: eps[i].normal_(), N * 4 * d),
        ("philox fill", lambda i: k.philox_normal(res, (B, K, d), dev), N * 4 * d),
        ("K15", lambda i: k.affine_propagate(x_prev[i], eps[i], y, *terms, scales, out_x=out_x[i]), N * (12 * d + 4)),
        ("K15 through ancestors", lambda i: k.affine_propagate(x_prev[i], eps[i], y, *terms, scales, out_x=out_x[i],
                                                               ancestors=idx[i]), N * (12 * d + 12)),
        ("K16 gather + noise inside", lambda i: k.affine_propagate_drawn(x_prev[i], res, y, *terms, scales, out_x=out_x[i],
                                                                         ancestors=idx[i]), N * (8 * d + 12)),
        ("K16 noise inside, no gather", lambda i: k.affine_propagate_drawn(x_prev[i], res, y, *terms, scales,
                                                                           out_x=out_x[i]), N * (8 * d + 4))]
print("B={} K={} d={}".format(B, K, d))
for name, fn, nbytes in rows:
    us = timeit(fn)
    print("{:48s} {:8.1f} us  {:7.1f} MB  {:6.2f} TB/s  {:.3f} of 8 TB/s".format(name, us, nbytes / 1e6, nbytes / us / 1e6,
                                                                          nbytes / us / 1e6 / 8))
