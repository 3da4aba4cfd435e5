"""Does a hipGraph replay of the north-star ELBO slow down with the SIZE OF THE CAPTURE'S MEMORY POOL?  The same
forward ELBO captured twice per batch size: with the autograd graph recorded (every timestep's tensors stay in the
pool) and with parameters that need no gradient (temporaries are reused).  Prints ms per replay and the pool size.

    python tools/graph_probe.py [B ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402
from aesmc_amd import graphs  # noqa: E402
from aesmc_amd.testing.models import LgssmNd  # noqa: E402


def main(batches):
    dev = torch.device("cuda", 0)
    for B in batches:
        for grad in (True, False):
            model = LgssmNd(10, dtype=torch.float32, affine=True, validate_args=False).tune_proposal().to(dev)
            for p in model.parameters():
                p.requires_grad_(grad)
            observations = model.simulate(100, B, seed=1)
            np.random.seed(0)
            torch.manual_seed(0)
            torch.cuda.synchronize()
            before = torch.cuda.memory_reserved()
            graphed = graphs.GraphedLoss(observations, 4096, "aesmc", model.initial, model.transition, model.emission,
                                         model.proposal, backward=False)
            for _ in range(2):
                graphed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                graphed()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print("B=%d autograd graph %s: %.2f ms per replay, pool %.1f GB" % (
                B, "recorded" if grad else "none", ms, (torch.cuda.memory_reserved() - before) / 2**30), flush=True)
            del graphed, model, observations
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [320, 384, 512])
