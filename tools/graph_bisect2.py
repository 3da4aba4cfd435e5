import subprocess
import sys

CASE = r'''
import sys, torch, numpy as np
sys.path.insert(0, ".")
from aesmc_amd import graphs, losses
from aesmc_amd.testing import models
dtype = torch.float64
model = models.LgssmNd(3, seed=0, dtype=dtype, validate_args=False).to("cuda")
obs = model.simulate(6, 8, seed=1)
parts = (model.initial, model.transition, model.emission, model.proposal)
variant = "{variant}"
if variant in ("eager_fwd", "eager_fwd_bwd", "eager_fwd_bwd_seed"):
    loss = losses.get_loss(obs, 64, "aesmc", *parts)
    if variant != "eager_fwd":
        loss.backward()
        model.zero_grad(set_to_none=True)
if variant == "eager_fwd_bwd_seed":
    torch.manual_seed(11); np.random.seed(11)
if variant == "seed_only":
    torch.manual_seed(11); np.random.seed(11)
g = graphs.GraphedLoss(obs, 64, "aesmc", *parts, backward=True)
print("captured", float(g()), float(g()))
'''
for variant in ["none", "seed_only", "eager_fwd", "eager_fwd_bwd", "eager_fwd_bwd_seed"]:
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CASE.format(variant=variant)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l or "File" in l][:6]
    print(variant, "rc", r.returncode, tail, err, flush=True)
