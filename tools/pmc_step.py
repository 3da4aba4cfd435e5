"""Workload for the hardware-counter passes over the fused resampling step (rocprofv3 --pmc, one set
of counters per run): the step on N(0,1) log-weights at configs[1]'s shape and at the 8-GPU shard's
shape, with one and with two workgroups per batch row, 5 launches each, in this order:
  c2 parts=1, c2 parts=2, c4s parts=1, c4s parts=2, c4 parts=1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels

k = _kernels.get()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
for (B, K, d), parts_list in (((256, 1024, 10), (1, 2)), ((128, 4096, 10), (1, 2)), ((1024, 4096, 10), (1,))):
    lw = torch.randn(B, K, device=dev, generator=gen)
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    x = torch.randn(B, K, d, device=dev, generator=gen)
    for parts in parts_list:
        k._lib.aesmc_test_set_step_parts(parts)
        for _ in range(5):
            k.resample_step(lw, u, x, want_lse=True)
        torch.cuda.synchronize()
k._lib.aesmc_test_set_step_parts(0)
print("done")
