"""K16 (item form) hipGraph-timed back to back at c4's shape (B = 1024, K = 4096, d = 10, through ancestors, six operand
sets), with and without the three densities' constants behind the weight pairs (aesmc_affine_weight_pairs_scaled against
aesmc_affine_weight_pairs: the launch then takes three logarithms per wavefront).  In the workload the same switch is
AESMC_MEASUREMENT_KNOBS=1 AESMC_K16_SCALED=0 python bench.py --workload c4 ...   (profiles/r06_k16_constants_ab.txt)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa
from aesmc_amd import _kernels, _ops, _philox
B, K, d = 1024, 4096, 10
dev = torch.device("cuda", 0)
k = _kernels.get()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)
SETS = 6
x_prev = [r(B, K, d) for _ in range(SETS)]
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

def timeit(fn, replays=8):
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

fn = lambda i: k.affine_propagate_drawn(x_prev[i], res, y, *terms, scales, out_x=out_x[i], ancestors=idx[i])
for trip in range(3):
    for scaled in (False, True):
        k.SCALED_PAIRS = scaled
        k.begin_evaluation()
        us = timeit(fn)
        tag = int(k._pairs[1][-2:-1].view(torch.int32).item()) if k._pairs[1].numel() % 2 == 0 else -1
        print("trip {} scaled={} tag={:#x}: {:.2f} us".format(trip, scaled, tag, us), flush=True)
