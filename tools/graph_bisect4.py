import subprocess
import sys

CASE = r'''
import sys, torch, numpy as np, gc
sys.path.insert(0, ".")
from aesmc_amd import graphs, losses
from aesmc_amd.testing import models
dtype = torch.float64
model = models.LgssmNd(3, seed=0, dtype=dtype, validate_args=False).to("cuda")
obs = model.simulate(6, 8, seed=1)
parts = (model.initial, model.transition, model.emission, model.proposal)
variant = "{variant}"
loss = losses.get_loss(obs, 64, "aesmc", *parts)
if variant == "del":
    del loss; gc.collect()
elif variant == "bwd_then_keep":
    loss.backward(); model.zero_grad(set_to_none=True)
elif variant == "bwd_then_del":
    loss.backward(); model.zero_grad(set_to_none=True); del loss; gc.collect()
g = graphs.GraphedLoss(obs, 64, "aesmc", *parts, backward=True)
print("captured", float(g()), float(g()))
'''
for variant in ["keep", "del", "bwd_then_keep", "bwd_then_del"]:
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CASE.format(variant=variant)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    print(variant, "rc", r.returncode, tail, flush=True)
