"""The forms of the fused propagation launch (K16) side by side at given shapes: persistent workgroups in two roles
(linear_gaussian_fused.hip) against one item per workgroup (linear_gaussian_item.hip), with K2 and the noise-fill + K15
pair for scale.  hipGraph-timed on six operand sets, N(0,1) operands, healthy ancestries.
    python tools/k16forms.py 128,4096,10 256,4096,10 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops, _philox  # noqa: E402

dev = torch.device("cuda", 0)
k = _kernels.get()
type(k).DRAWN_MIN_PARTICLES = 0
lib = k._lib
SETS = 6


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


def shape(B, K, d):
    gen = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)
    x_prev = [r(B, K, d) for _ in range(SETS)]
    eps = [r(B, K, d) for _ in range(SETS)]
    out_x = [torch.empty(B, K, d, device=dev) for _ in range(SETS)]
    lw = [r(B, K) for _ in range(SETS)]
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    y = r(B, d)
    eye = torch.eye(d, device=dev)
    A, C, Q = 0.9 * eye + 0.01 * r(d, d), eye + 0.01 * r(d, d), 0.45 * eye + 0.01 * r(d, d)
    terms = ((A, None), (C, None), (Q, r(B, d)))
    scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
    idx = [_ops.ancestor_index(w, u) for w in lw]
    res = _philox.reserve(B * K * d, dev)
    N = B * K
    drawn = lambda i: k.affine_propagate_drawn(x_prev[i], res, y, *terms, scales, out_x=out_x[i], ancestors=idx[i])
    rows = [("K2 (indices + lse + ranges)", lambda i: k.resample_step(lw[i], u, None, True, want_child_end=True), N * 16, 0),
            ("noise fill", lambda i: k.philox_normal(res, (B, K, d), dev), N * 4 * d, 0),
            ("K15 through ancestors", lambda i: k.affine_propagate(x_prev[i], eps[i], y, *terms, scales, out_x=out_x[i],
                                                                   ancestors=idx[i]), N * (12 * d + 12), 0),
            ("K16 persistent form", drawn, N * (8 * d + 12), 1), ("K16 item form, v_fmac chains", drawn, N * (8 * d + 12), 2),
            ("K16 item form, v_pk_fma pairs", drawn, N * (8 * d + 12), 3)]
    print("B={} K={} d={}".format(B, K, d), flush=True)
    for name, fn, nbytes, form in rows:
        k.WEIGHT_PAIRS = form == 3
        form = min(form, 2)
        lib.aesmc_test_set_k16_form(form)
        if form and fn(0) is None:
            print("{:32s} declined".format(name))
            continue
        ran = lib.aesmc_test_last_k16_form() if form else 0
        us = timeit(fn)
        print("{:32s} {:8.1f} us  {:7.1f} MB  {:6.2f} TB/s  {:.3f} of 8 TB/s{}".format(
            name, us, nbytes / 1e6, nbytes / us / 1e6, nbytes / us / 1e6 / 8,
            "" if not form or ran == form else "   (ran as form {})".format(ran)), flush=True)
    lib.aesmc_test_set_k16_form(0)
    k.WEIGHT_PAIRS = True


for spec in sys.argv[1:] or ["128,4096,10", "256,4096,10", "512,4096,10", "1024,4096,10"]:
    shape(*[int(v) for v in spec.split(",")])
