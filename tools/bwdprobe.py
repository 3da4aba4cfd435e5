"""TEMPORARY: where the range backward kernel spends its time on collapsed indices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesmc_amd import _kernels
from tools.stepbench import timeit
k = _kernels.get()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
for (B, K, d) in [(1024, 4096, 10), (256, 1024, 10)]:
    x = torch.randn(B, K, d, device=dev, generator=gen)
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    for s in (1.0, 5.0):
        idx = k.ancestor_index(s * torch.randn(B, K, device=dev, generator=gen), u)
        runs = (idx[:, 1:] != idx[:, :-1]).sum().item() + B
        for which, label in ((1, "old + zero fill"), (0, "range"), (2, "range, no output stage"), (3, "range, no sums"), (4, "range, no look-back")):
            k._lib.aesmc_set_sorted_backward_kernel(which)
            t = timeit(lambda: k.gather_backward(x, idx, sorted_index=True))
            print((B, K, d), "s=%g unique=%.3f" % (s, runs / (B * K)), label, "%.1f us" % t)
        k._lib.aesmc_set_sorted_backward_kernel(0)
