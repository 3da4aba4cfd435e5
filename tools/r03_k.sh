#!/bin/bash
OUT=gpurun_out
for case in "aesmc-False" "iwae-False" "aesmc-True"; do
  echo "== $case"; timeout -k 10 300 python -m pytest "tests/test_gpu_graphs.py::test_graphed_forward_backward_equals_eager[$case]" -q -x 2>&1 | tail -3 | cut -c1-200
done
echo "== lazy off"; AESMC_LAZY_GATHER=0 timeout -k 10 300 python -m pytest "tests/test_gpu_graphs.py::test_graphed_forward_backward_equals_eager[aesmc-False]" -q -x 2>&1 | tail -2 | cut -c1-200
echo "== forward-only capture"; timeout -k 10 300 python - <<'PY' 2>&1 | tail -5
import torch, numpy as np
import aesmc_amd
from aesmc_amd import graphs, losses
from aesmc_amd.testing import models
dev = torch.device("cuda", 0)
model = models.LgssmNd(3, seed=0, dtype=torch.float64, validate_args=False, affine=False).to(dev)
obs = model.simulate(6, 8, seed=1)
parts = (model.initial, model.transition, model.emission, model.proposal)
g = graphs.GraphedLoss(obs, 64, "aesmc", *parts, backward=False)
print("forward capture ok", float(g()))
g = graphs.GraphedLoss(obs, 64, "aesmc", *parts, backward=True, verify_replays=0)
print("backward capture ok", float(g()))
PY
