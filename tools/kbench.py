"""Kernel micro-benchmark: every C-ABI kernel at the BASELINE.json shapes, launched back to back
between two HIP events (same method as bench.py's roofline leg).  Prints us per launch and GB/s of
algorithmic bytes.  Usage: python tools/kbench.py [c2 c4 c5 c3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels

SHAPES = {"c2": (256, 1024, 10), "c4": (1024, 4096, 10), "c4s": (128, 4096, 10), "c5": (64, 16384, 128),
          "c3": (4096, 8192, 1)}


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main(names):
    k = _kernels.get()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in names:
        B, K, d = SHAPES[name]
        print("== {} B={} K={} d={}".format(name, B, K, d))
        rows = []
        a, b, c = [torch.randn(B, K, device=dev, generator=gen) for _ in range(3)]
        us = timeit(lambda: k.logweight_lse(a, b, c))
        rows.append(("K1 logweight_lse", us, B * K * 16 + 4 * B))
        lw, lse = k.logweight_lse(a, b, c)
        gl = torch.randn(B, device=dev, generator=gen)
        us = timeit(lambda: k.logweight_lse_backward(lw, lse, None, gl))
        rows.append(("K1 backward", us, B * K * 12))
        u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
        us = timeit(lambda: k.ancestor_index(lw, u))
        rows.append(("K2 ancestor_index s=1", us, B * K * 12 + 8 * B))
        lw5 = 5 * lw
        us = timeit(lambda: k.ancestor_index(lw5, u))
        rows.append(("K2 ancestor_index s=5", us, B * K * 12 + 8 * B))
        idx = k.ancestor_index(lw, u)
        idx5 = k.ancestor_index(lw5, u)
        if name != "c3":
            x = torch.randn(B, K, d, device=dev, generator=gen)
            loc = torch.randn(B, K, d, device=dev, generator=gen)
            us = timeit(lambda: k.gather(x, idx))
            rows.append(("K3 gather s=1", us, B * K * (8 + 8 * d)))
            us = timeit(lambda: k.gather(x, idx5))
            rows.append(("K3 gather s=5", us, B * K * (8 + 8 * d)))
            for which, label in ((1, "source tiles + zero fill"), (0, "range kernel, no zero fill")):
                k._lib.aesmc_test_set_sorted_backward_kernel(which)
                us = timeit(lambda: k.gather_backward(x, idx, sorted_index=True))
                rows.append(("K3 backward s=1 ({})".format(label), us, B * K * (8 + 8 * d)))
                us = timeit(lambda: k.gather_backward(x, idx5, sorted_index=True))
                rows.append(("K3 backward s=5 ({})".format(label), us, B * K * (8 + 8 * d)))
            us = timeit(lambda: k.gather_backward(x, idx))
            rows.append(("K3 backward s=1 (atomic path)", us, B * K * (8 + 8 * d)))
            scale = torch.tensor(0.7, device=dev).expand(B, K, d)
            us = timeit(lambda: k.normal_logprob_sum(x, loc, scale))
            rows.append(("K4 normal_logprob_sum", us, B * K * (8 * d + 4)))
            go = torch.randn(B, K, device=dev, generator=gen)
            us = timeit(lambda: k.normal_logprob_sum_backward(x, loc, scale, go, True, True, False))
            rows.append(("K4 backward (value, loc)", us, B * K * (16 * d + 4)))
            y = torch.randn(B, d, device=dev, generator=gen).unsqueeze(1).expand(B, K, d)
            # K5 and its backward in the layout of a Markov model's timestep: x, three dense locations,
            # the observation one row per batch element, scalar scales
            loc2, loc3 = [torch.randn(B, K, d, device=dev, generator=gen) for _ in range(2)]
            us = timeit(lambda: k.normal_logweight(x, loc, scale, y, loc2, scale, loc3, scale))
            rows.append(("K5 normal_logweight", us, B * K * (16 * d + 4)))
            lw5 = k.normal_logweight(x, loc, scale, y, loc2, scale, loc3, scale)
            _, lse5 = k.logweight_lse(lw5, None, None, want_lw=False)
            need = [True, True, False, False, True, False, True, False]
            us = timeit(lambda: k.normal_logweight_backward(x, loc, scale, y, loc2, scale, loc3, scale, go, need))
            rows.append(("K5 backward (x, 3 locs), grad_lw given", us, B * K * (32 * d + 4)))
            us = timeit(lambda: k.normal_logweight_backward(x, loc, scale, y, loc2, scale, loc3, scale, None, need,
                                                             lw=lw5, lse=lse5, grad_lse=gl))
            rows.append(("K5 backward fused with K1's (lse)", us, B * K * (32 * d + 4)))
            us = timeit(lambda: k.normal_logprob_sum(y, loc, scale))
            rows.append(("K4 (value = expanded obs)", us, B * K * (4 * d + 4)))
            us = timeit(lambda: x.clone())
            rows.append(("torch clone [B,K,d] (copy roof)", us, B * K * 8 * d))
        for label, us, nbytes in rows:
            print("  {:46s} {:9.2f} us  {:8.1f} GB/s".format(label, us, nbytes / us / 1e3))


if __name__ == "__main__":
    main(sys.argv[1:] or ["c2", "c4"])
