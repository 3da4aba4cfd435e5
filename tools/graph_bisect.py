"""Debug helper: tries GraphedLoss capture variants in subprocesses (a capture bug may segfault)."""
import subprocess
import sys

CASE = r'''
import sys, torch, numpy as np
sys.path.insert(0, ".")
from aesmc_amd import graphs
from aesmc_amd.testing import models
dtype = getattr(torch, "{dtype}")
model = models.LgssmNd({d}, seed=0, dtype=dtype, validate_args=False).to("cuda")
obs = model.simulate({T}, {B}, seed=1)
g = graphs.GraphedLoss(obs, {K}, "{alg}", model.initial, model.transition, model.emission, model.proposal, backward={bwd})
print("captured", float(g()), float(g()))
'''

cases = [dict(dtype="float64", d=3, T=6, B=8, K=64, alg="aesmc", bwd=True),
         dict(dtype="float32", d=3, T=6, B=8, K=64, alg="aesmc", bwd=True),
         dict(dtype="float64", d=3, T=6, B=8, K=64, alg="aesmc", bwd=False),
         dict(dtype="float64", d=3, T=6, B=8, K=64, alg="iwae", bwd=True),
         dict(dtype="float64", d=10, T=6, B=64, K=1024, alg="aesmc", bwd=True),
         dict(dtype="float32", d=10, T=6, B=8, K=64, alg="aesmc", bwd=True),
         dict(dtype="float32", d=3, T=2, B=8, K=64, alg="aesmc", bwd=True),
         dict(dtype="float32", d=10, T=50, B=256, K=1024, alg="aesmc", bwd=True)]
for case in cases:
    r = subprocess.run([sys.executable, "-c", CASE.format(**case)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l][:2]
    print(case, "rc", r.returncode, tail, err, flush=True)
