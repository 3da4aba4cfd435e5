"""K7 (particle summaries) at the BASELINE.json shapes, hipGraph-timed, beside the PyTorch expression."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels
from tools.stepbench import timeit

k = _kernels.get()
dev = torch.device("cuda", 0)
for (B, K, d) in [(256, 1024, 10), (1024, 4096, 10), (128, 4096, 10), (64, 16384, 128), (4096, 8192, 1), (4, 32768, 10)]:
    lw = torch.randn(B, K, device=dev)
    v = torch.randn(B, K, d, device=dev)
    t = timeit(lambda: k.particle_summary(lw, v, True, True, True))
    w = torch.softmax(lw, 1).unsqueeze(-1)
    t2 = timeit(lambda: ((w * v).sum(1), (w * v * v).sum(1)))
    print((B, K, d), "K7 %.1f us  %.0f GB/s | torch mean+second given weights %.1f us" % (
        t, (v.numel() + lw.numel()) * 4 / t / 1e3, t2))
