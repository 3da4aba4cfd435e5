"""The two kernels of a payload-free resampling step side by side (general: ancestor_index_inv_kernel; rows: the lean
form), with and without the children ranges, hipGraph-timed on six operand sets of N(0,1) log-weights.
    python tools/k2forms.py 1024,4096 128,4096 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels  # noqa: E402

dev = torch.device("cuda", 0)
k = _kernels.get()
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


for spec in sys.argv[1:] or ["1024,4096", "512,4096", "256,4096", "128,4096", "256,1024", "64,16384"]:
    B, K = [int(v) for v in spec.split(",")]
    gen = torch.Generator(device=dev).manual_seed(0)
    lw = [torch.randn(B, K, device=dev, generator=gen) for _ in range(SETS)]
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    print("B={} K={}".format(B, K), flush=True)
    for ranges in (True, False):
        for form, name in ((1, "general"), (2, "rows")):
            lib.aesmc_test_set_k2_form(form)
            k.resample_step(lw[0], u, None, True, want_child_end=ranges)
            ran = lib.aesmc_test_last_k2_form()
            us = timeit(lambda i: k.resample_step(lw[i], u, None, True, want_child_end=ranges))
            nbytes = B * K * (16 if ranges else 12)
            print("  {:8s} {:14s} {:7.1f} us  {:6.1f} MB  {:5.2f} TB/s  {:.3f} of 8 TB/s{}".format(
                name, "with ranges" if ranges else "indices only", us, nbytes / 1e6, nbytes / us / 1e6, nbytes / us / 8e6,
                "" if ran == form else "  (ran as {})".format(ran)), flush=True)
    lib.aesmc_test_set_k2_form(0)
