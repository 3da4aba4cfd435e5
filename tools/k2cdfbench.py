"""The resampling launch as the timed path issues it (indices + logsumexp + children ranges, no payload) under both
float32-CDF modes (aesmc_set_float32_cdf) at the BASELINE.json shapes.  Timing as tools/stepbench.py: calls captured
in one hipGraph and replayed.  Usage: python tools/k2cdfbench.py [c2 c4 c5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels, inference
from tools.kbench import SHAPES
from tools.stepbench import timeit


def main(names):
    k = _kernels.get()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in names:
        B, K, d = SHAPES[name]
        for s in (1.0, 5.0):
            lw = s * torch.randn(B, K, device=dev, generator=gen)
            u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
            line = "{} B={} K={} s={}:".format(name, B, K, s)
            got = {}
            for mode in ("float64", "reference"):
                previous = inference.set_float32_cdf(mode)
                try:
                    t_step = timeit(lambda: k.resample_step(lw, u, None, want_lse=True, want_child_end=True))
                    t_k2 = timeit(lambda: k.ancestor_index(lw, u))
                    got[mode] = k.ancestor_index(lw, u)
                finally:
                    inference.set_float32_cdf(previous)
                line += "  [{}] step {:.2f} us, indices alone {:.2f} us".format(mode, t_step, t_k2)
            differ = int((got["float64"] != got["reference"]).sum())
            print(line + "  | indices that differ between the modes: {} of {}".format(differ, B * K), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["c2", "c4", "c5"])
