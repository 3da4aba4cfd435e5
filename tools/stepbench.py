"""Fused resampling step vs its parts at the BASELINE.json shapes (same timing method as
tools/kbench.py).  Usage: python tools/stepbench.py [c2 c4 c4s c5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels
from tools.kbench import SHAPES


def timeit(fn, reps=20, replays=5):
    """Device time per call: `reps` calls captured in one hipGraph, replayed (no host gaps)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
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


def main(names):
    k = _kernels.get()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in names:
        B, K, d = SHAPES[name]
        print("== {} B={} K={} d={}".format(name, B, K, d))
        for s in (1.0, 5.0):
            lw = s * torch.randn(B, K, device=dev, generator=gen)
            u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
            x = torch.randn(B, K, d, device=dev, generator=gen)
            t_k1 = timeit(lambda: k.logweight_lse(lw, None, None, want_lw=False, want_lse=True))
            t_k2 = timeit(lambda: k.ancestor_index(lw, u))
            idx = k.ancestor_index(lw, u)
            t_k3 = timeit(lambda: k.gather(x, idx))
            t_idx_lse = timeit(lambda: k.resample_step(lw, u, None, want_lse=True))
            t_fused = timeit(lambda: k.resample_step(lw, u, x, want_lse=True))
            print("  s={}: K1 {:.2f}  K2 {:.2f}  K3 {:.2f}  sum {:.2f} | step(idx+lse) {:.2f}  "
                  "step(idx+lse+gather) {:.2f} us".format(s, t_k1, t_k2, t_k3, t_k1 + t_k2 + t_k3, t_idx_lse,
                                                          t_fused))


if __name__ == "__main__":
    main(sys.argv[1:] or ["c2", "c4"])
