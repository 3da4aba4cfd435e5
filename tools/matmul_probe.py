"""Micro-probe: how fast are the skinny [B*K, d] x [d, d] maps of SSM callables on this box?"""
import os
import sys
import time

import torch

def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for (B, K, d) in [(256, 1024, 10), (1024, 4096, 10)]:
    x = torch.randn(B, K, d, device="cuda")
    W = torch.randn(d, d, device="cuda")
    b = torch.randn(d, device="cuda")
    mb = 2 * x.numel() * 4 / 1e6
    res = {}
    res["x @ W.t()"] = bench(lambda: x @ W.t())
    res["F.linear"] = bench(lambda: torch.nn.functional.linear(x, W, b))
    res["einsum"] = bench(lambda: torch.einsum("bkd,ed->bke", x, W))
    res["copy (x*2)"] = bench(lambda: x * 2)
    try:
        torch.backends.cuda.preferred_blas_library("hipblas")
        res["rocblas x@W.t()"] = bench(lambda: x @ W.t())
        torch.backends.cuda.preferred_blas_library("hipblaslt")
    except Exception as e:
        res["rocblas"] = str(e)
    print((B, K, d), "MB moved %.0f" % mb, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}, flush=True)
