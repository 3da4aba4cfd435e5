"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per
run): launches K3 on known operands so HBM traffic per launch can be read off the counters.

Order of resample_gather launches (3 each): for shape in (c2, c4):
  calibration: identity index (every source row read exactly once: bytes known exactly)
  workload   : systematic-resampling indices from log-weights ~ N(0,1)   (ESS/K ~ 0.37)
  degenerate : indices from log-weights ~ 5 N(0,1)                        (few survivors)
Order of ancestor_index_inv_kernel launches per shape: K2 alone on the two weight sets (1 each),
then the fused step (K2 + row log-sum-exp + payload gather) 3x on each weight set.
Then 3 launches of K5 (normal_logweight_kernel) and 3 of K6 (normal_rsample_dense_kernel) per shape.
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesmc_amd import _kernels

k = _kernels.get()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
for (B, K, d) in [(256, 1024, 10), (1024, 4096, 10)]:
    value = torch.randn(B, K, d, device=dev, generator=gen)
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
    identity = torch.arange(K, device=dev).unsqueeze(0).expand(B, K).contiguous()
    lw1 = torch.randn(B, K, device=dev, generator=gen)
    lw5 = 5 * torch.randn(B, K, device=dev, generator=gen)
    idx1 = k.ancestor_index(lw1, u)
    idx5 = k.ancestor_index(lw5, u)
    torch.cuda.synchronize()
    for idx in (identity, idx1, idx5):
        for _ in range(3):
            k.gather(value, idx)
        torch.cuda.synchronize()
    for lw in (lw1, lw5):
        for _ in range(3):
            k.resample_step(lw, u, value, want_lse=True)
        torch.cuda.synchronize()
    # K5 and K6 on operands of the same shape (3 launches each): traffic against algorithmic bytes
    loc_p, loc_g, loc_q, eps = [torch.randn(B, K, d, device=dev, generator=gen) for _ in range(4)]
    obs = torch.randn(B, d, device=dev, generator=gen).unsqueeze(1).expand(B, K, d)
    scale = torch.tensor(0.7, device=dev).expand(B, K, d)
    for _ in range(3):
        k.normal_logweight(value, loc_p, scale, obs, loc_g, scale, loc_q, scale)
    torch.cuda.synchronize()
    for _ in range(3):
        k.normal_rsample(eps, loc_q, scale)
    torch.cuda.synchronize()
    uniq1 = float((idx1[:, 1:] != idx1[:, :-1]).sum() + B) / (B * K)
    uniq5 = float((idx5[:, 1:] != idx5[:, :-1]).sum() + B) / (B * K)
    print("shape", (B, K, d), "unique-ancestor fraction: s=1 %.3f, s=5 %.3f" % (uniq1, uniq5))
