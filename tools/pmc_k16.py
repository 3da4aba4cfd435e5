"""Workload for hardware-counter passes over the forward step's launches at B=1024 K=4096 d=10 (rocprofv3 --pmc):
K2, K15, K15 through the ancestors, K16; 3 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops, _philox  # noqa: E402

B, K, d = [int(v) for v in os.environ.get("K16_SHAPE", "1024,4096,10").split(",")]
dev = torch.device("cuda", 0)
k = _kernels.get()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)
x_prev, eps, out_x, lw = r(B, K, d), r(B, K, d), torch.empty(B, K, d, device=dev), r(B, K)
u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
y = r(B, d)
eye = torch.eye(d, device=dev)
terms = ((0.9 * eye + 0.01 * r(d, d), None), (eye + 0.01 * r(d, d), None), (0.45 * eye + 0.01 * r(d, d), r(B, d)))
scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
idx = _ops.ancestor_index(lw, u)
res = _philox.reserve(B * K * d, dev)
for _ in range(3):
    k.resample_step(lw, u, None, True)
    k.philox_normal(res, (B, K, d), dev)
    k.affine_propagate(x_prev, eps, y, *terms, scales, out_x=out_x)
    k.affine_propagate(x_prev, eps, y, *terms, scales, out_x=out_x, ancestors=idx)
    k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=out_x, ancestors=idx)
torch.cuda.synchronize()
print("done")
