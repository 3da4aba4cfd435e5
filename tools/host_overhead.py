"""Host cost of one timestep of the eager ELBO loop: the headline model at a size whose kernels are
negligible (B=8, K=64), timed and profiled (cProfile) on the GPU box.

    python tools/host_overhead.py [--grad 1] [--affine 0]      (--affine 0: callables in the reference's own style,
                                                                 Normal(x @ W.t() + c, s), recorded on the lazy latents)
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402
from aesmc_amd import losses  # noqa: E402
from aesmc_amd.testing.models import LgssmNd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grad", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--affine", type=int, default=1)
    ap.add_argument("--particles", type=int, default=64, help="K (>= 128: the steps take K16 as at the BASELINE shapes)")
    ap.add_argument("--profile", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    T = 100
    model = LgssmNd(10, dtype=torch.float32, affine=bool(args.affine), validate_args=False).tune_proposal().to(dev)
    observations = model.simulate(T, 8, seed=1)
    np.random.seed(0)
    torch.manual_seed(0)

    def step():
        with torch.set_grad_enabled(bool(args.grad)):
            return losses.get_loss(observations, args.particles, "aesmc", model.initial, model.transition, model.emission,
                                   model.proposal)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("host us per timestep: %.1f" % (1e6 * dt / (args.steps * T)))
    if not args.profile:
        return
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    prof.disable()
    stats = pstats.Stats(prof)
    stats.sort_stats("cumulative").print_stats(60)
    stats.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
