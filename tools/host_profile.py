"""cProfile of the eager timestep loop at configs[1] sizes: where does the host spend its time?"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import aesmc_amd
from aesmc_amd.testing import models

device = torch.device("cuda", 0)
model = models.LgssmNd(10, seed=0, validate_args=False).to(device)
obs = model.simulate(50, 256, seed=1)
parts = (model.initial, model.transition, model.emission, model.proposal)
for _ in range(3):
    aesmc_amd.losses.get_loss(obs, 1024, "aesmc", *parts)
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(5):
    aesmc_amd.losses.get_loss(obs, 1024, "aesmc", *parts)
torch.cuda.synchronize()
prof.disable()
stats = pstats.Stats(prof)
stats.sort_stats("tottime").print_stats(28)
