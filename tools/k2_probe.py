import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesmc_amd import _kernels
k = _kernels.get()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
for (B, K) in [(1024, 4096), (256, 1024)]:
    lw = torch.randn(B, K, device=dev, generator=gen)
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    for _ in range(3):
        k.ancestor_index(lw, u)
    torch.cuda.synchronize()
