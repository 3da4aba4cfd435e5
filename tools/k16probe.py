"""Times the fused propagation launch (K16) with parts of it switched off (a build with -DAESMC_K16_PROBES; the probed
launches' OUTPUT IS WRONG): which part of the launch the time belongs to.  hipGraph-timed on cycled operand sets.
NOTE (round 6): the probe / stamp code this script drives was removed from the product translation units (VERDICT r05,
hygiene).  The instrumented kernels are the tree at commit 549e68a: to repeat the experiment, check that commit's
aesmc_amd/csrc/ out into tools/exp/ (git-ignored), build it with AESMC_HIPCC_FLAGS=-DAESMC_K16_PROBES (or -DAESMC_K14_PROBES)
and AESMC_PROBE_BUILD=1, and point the loader at that library.  The results it produced are under profiles/ (r04_k16_stamps.txt,
r04_k14_probes.txt, r05_pmc_k14_c4.txt, ...).
"""
import os
os.environ.setdefault("AESMC_MEASUREMENT_KNOBS", "1")      # the library reads AESMC_* knobs only beside this
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops, _philox  # noqa: E402

B, K, d = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 4096, 10))]
dev = torch.device("cuda", 0)
k = _kernels.get()
type(k).DRAWN_MIN_PARTICLES = 0
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


NAMES = {1: "no draws", 2: "no particle arithmetic", 4: "no x_t stores", 8: "no row loads", 16: "no ancestor loads",
         32: "no emission part"}
print("B={} K={} d={}".format(B, K, d))
masks = [int(v) for v in sys.argv[4:]] or [0, 1, 2, 3, 4, 8, 16, 24, 28, 30, 31, 32, 0]
for mask in masks:
    os.environ["AESMC_K16_PROBE"] = str(mask)
    us = timeit(lambda i: k.affine_propagate_drawn(x_prev[i], res, y, *terms, scales, out_x=out_x[i], ancestors=idx[i]))
    what = " + ".join(NAMES[b] for b in sorted(NAMES) if mask & b) or "the product launch"
    print("probe {:3d}  {:8.1f} us   {}".format(mask, us, what))
