"""ms per timestep of one SMC forward (no autograd) at B K = 2^20 particles for a range of latent widths d (dx = dy = d):
the library's fused route for that width — the item kernels K16 up to 16, the matrix-core step K17g / K18g (K17 / K18 at
128) from 20 to 256 — against the GENERIC route (the model's callables written with PyTorch matmuls, lazy latents off:
three library GEMMs, their offset adds, K6 draw, K5 log-weight, K2, K3 per timestep).  VERDICT r05 item 2:
"no width in 2 ... 256 slower than the generic route".

    python tools/width_sweep.py [--widths 8,16,24,...] [--batch 64] [--particles 16384] [--steps 10]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import aesmc_amd  # noqa: E402
from aesmc_amd import _kernels, inference  # noqa: E402
from aesmc_amd.testing import models  # noqa: E402


def timed(fn, repeats=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        fn()
        stop.record()
        torch.cuda.synchronize()
        best = min(best, start.elapsed_time(stop))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--widths", default="8,16,20,24,32,48,64,96,128,192,256")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--particles", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    provider = _kernels.get()
    B, K, T = args.batch, args.particles, args.steps
    print("# B = {} K = {} (B K = {}), T = {}; ms per timestep = (time of T steps - time of 2 steps) / (T - 2): a "
          "resampled timestep, the first step's proposal aside; best of 3; {}".format(
              B, K, B * K, T, torch.cuda.get_device_name(0)))
    print("width  fused_ms  generic_ms  speedup  fused_route")
    for d in [int(v) for v in args.widths.split(",")]:
        rows = {}
        for label, affine, lazy in (("fused", True, True), ("generic", False, False)):
            model = models.LgssmNd(d, seed=0, validate_args=False, affine=affine).to(device).tune_proposal()
            observations = model.simulate(T, B, seed=1)
            parts = (model.initial, model.transition, model.emission, model.proposal)
            calls = {"wide": 0, "item": 0}
            real_wide, real_item = provider.affine_propagate_wide, provider.affine_propagate_drawn

            def wide(*a, **k):
                out = real_wide(*a, **k)
                calls["wide"] += out is not None
                return out

            def item(*a, **k):
                out = real_item(*a, **k)
                calls["item"] += out is not None
                return out
            provider.affine_propagate_wide, provider.affine_propagate_drawn = wide, item

            def run(steps):
                torch.manual_seed(0)
                np.random.seed(0)
                with torch.no_grad(), inference.lazy_gather(lazy):
                    return inference.infer("smc", observations[:steps], *parts, K, return_log_marginal_likelihood=True,
                                           return_latents=False, return_log_weight=False)
            try:
                full = timed(lambda: run(T))
                short = timed(lambda: run(2))
            finally:
                provider.affine_propagate_wide, provider.affine_propagate_drawn = real_wide, real_item
            rows[label] = ((full - short) / (T - 2), "matrix cores" if calls["wide"] else ("item kernels" if calls["item"] else "none"))
            del model, observations
            torch.cuda.empty_cache()
        fused, generic = rows["fused"][0], rows["generic"][0]
        print("{:5d}  {:8.3f}  {:10.3f}  {:7.2f}  {}".format(d, fused, generic, generic / fused, rows["fused"][1]))
    _kernels.get().read_flags(device)


if __name__ == "__main__":
    main()
