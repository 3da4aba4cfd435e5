"""Fused resampling step vs its parts at the BASELINE.json shapes, and the step for every number
of workgroups per batch row (`aesmc_test_set_step_parts`).  Timing: calls captured in one hipGraph and
replayed (device time, no host gaps).  Usage: python tools/stepbench.py [c2 c4s c4 c5]"""
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
    lib = k._lib
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in names:
        B, K, d = SHAPES[name]
        print("== {} B={} K={} d={}".format(name, B, K, d))
        for s in (1.0, 5.0):
            lw = s * torch.randn(B, K, device=dev, generator=gen)
            u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
            x = torch.randn(B, K, d, device=dev, generator=gen)
            lib.aesmc_test_set_step_parts(0)
            t_k1 = timeit(lambda: k.logweight_lse(lw, None, None, want_lw=False, want_lse=True))
            t_k2 = timeit(lambda: k.ancestor_index(lw, u))
            idx = k.ancestor_index(lw, u)
            t_k3 = timeit(lambda: k.gather(x, idx))
            t_idx_lse = timeit(lambda: k.resample_step(lw, u, None, want_lse=True))
            unique = (int((idx[:, 1:] != idx[:, :-1]).sum()) + B) / (B * K)
            print("  s={} unique={:.3f}: K1 {:.2f}  K2 {:.2f}  K3 {:.2f}  sum {:.2f} | step(idx+lse) {:.2f} us".format(
                s, unique, t_k1, t_k2, t_k3, t_k1 + t_k2 + t_k3, t_idx_lse))
            if k.resample_step(lw, u, x, want_lse=True) is None:
                print("    fused step declines this payload")
                continue
            algorithmic = B * K * (20 + 8 * d) + 8 * B
            moved = B * K * 12 + (1 + unique) * B * K * d * 4
            want = k.resample_step(lw, u, x, want_lse=True)
            for parts in (0, 1, 2, 4):
                lib.aesmc_test_set_step_parts(parts)
                got = k.resample_step(lw, u, x, want_lse=True)
                same = all(torch.equal(a, b) for a, b in zip(got, want))
                t = timeit(lambda: k.resample_step(lw, u, x, want_lse=True))
                print("    step(idx+lse+gather) parts={:>4} : {:7.2f} us  {:6.2f} TB/s algorithmic  {:6.2f} TB/s moved  same={}".format(
                    parts or "auto", t, algorithmic / t / 1e6, moved / t / 1e6, same))
            lib.aesmc_test_set_step_parts(0)


if __name__ == "__main__":
    main(sys.argv[1:] or ["c2", "c4s", "c4"])
