import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import aesmc_amd
from aesmc_amd import _kernels, _philox, _ops, state, inference
from aesmc_amd.testing.models import LgssmNd
dev = torch.device("cuda", 0)
k = _kernels.get()
B, K, d = 8, 128, 10
model = LgssmNd(d, dtype=torch.float32, affine=True, validate_args=False).tune_proposal().to(dev)
lw = torch.randn(B, K, device=dev); u = torch.rand(B, device=dev, dtype=torch.float64)
x = torch.randn(B, K, d, device=dev); y = torch.randn(B, d, device=dev); off = torch.randn(B, d, device=dev)
out_x = torch.empty_like(x)
scales = (model.transition_scale, model.emission_scale, model.proposal_scale)
terms = ((model.A.detach(), None), (model.C.detach(), None), (model.Wx.detach(), off))
def timeit(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t; torch.cuda.synchronize()
    return 1e6 * dt / n
idx = k.resample_step(lw, u, None, want_lse=True, want_child_end=True)[0]
print("K2 wrapper  (resample_step, ranges)      %.1f us" % timeit(lambda: k.resample_step(lw, u, None, want_lse=True, want_child_end=True)))
def k16():
    noise = _philox.reserve(B * K * d, dev)
    return k.affine_propagate_drawn(x, noise, y, *terms, scales, out_x, ancestors=idx)
print("reserve + K16 wrapper                    %.1f us" % timeit(k16))
print("philox reserve alone                     %.1f us" % timeit(lambda: _philox.reserve(B * K * d, dev)))
print("torch.empty [B,K,d]                      %.1f us" % timeit(lambda: torch.empty((B, K, d), device=dev)))
lib = k._lib
import ctypes
print("bare ctypes call aesmc_version           %.2f us" % timeit(lambda: lib.aesmc_version(), 20000))
from aesmc_amd.linear_gaussian import AffineNormal
print("AffineNormal ctor                        %.1f us" % timeit(lambda: AffineNormal(x, model.A, model.transition_scale, validate_args=False)))
print("torch Normal ctor (validate_args=False)  %.1f us" % timeit(lambda: torch.distributions.Normal(x, model.transition_scale, validate_args=False)))
obs = model.simulate(100, B, seed=1)
def full():
    return aesmc_amd.losses.get_loss(obs, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
print("get_loss per timestep (grad)             %.1f us" % (timeit(full, 20) / 100))
def full_ng():
    with torch.no_grad():
        return aesmc_amd.losses.get_loss(obs, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
print("get_loss per timestep (no grad)          %.1f us" % (timeit(full_ng, 20) / 100))
