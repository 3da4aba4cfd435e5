import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd
from aesmc_amd import losses
from aesmc_amd.testing.models import LgssmNd
dev = torch.device("cuda", 0)
model = LgssmNd(10, dtype=torch.float32, affine=True, validate_args=False).tune_proposal().to(dev)
B, K, T = 64, 4096, 12
obs = model.simulate(T, B, seed=1)
def step():
    model.zero_grad(set_to_none=True)
    loss = losses.get_loss(obs, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
for e in rows[:60]:
    dt = getattr(e, "device_time_total", None)
    if dt is None: dt = getattr(e, "cuda_time_total", 0)
    print("{:70s} n={:5d} cpu={:9.1f}us dev={:9.1f}us".format(e.key[:70], e.count, e.cpu_time_total, dt))
