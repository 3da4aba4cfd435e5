"""Timing experiments on the rows form of the step's backward (linear_gaussian_step_backward.hip) built with
-DAESMC_K14_PROBES: AESMC_K14_PROBE is a bit mask of pieces to LEAVE OUT (1 children's sum, 2 the three locations,
4 the three adjoints, 8 the matrix-core outer products and their LDS reads, 16 the LDS stores in front of them,
32 the gradient's stores, 64 the next tile's row loads).  A probed launch's output is wrong; only its time means anything.

    python tools/k14probe.py [mask ...]

NOTE (round 6): the probe / stamp code this script drives was removed from the product translation units (VERDICT r05,
hygiene).  The instrumented kernels are the tree at commit 549e68a: to repeat the experiment, check that commit's
aesmc_amd/csrc/ out into tools/exp/ (git-ignored), build it with AESMC_HIPCC_FLAGS=-DAESMC_K16_PROBES (or -DAESMC_K14_PROBES)
and AESMC_PROBE_BUILD=1, and point the loader at that library.  The results it produced are under profiles/ (r04_k16_stamps.txt,
r04_k14_probes.txt, r05_pmc_k14_c4.txt, ...).
"""
import os
os.environ.setdefault("AESMC_MEASUREMENT_KNOBS", "1")      # the library reads AESMC_* knobs only beside this
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels  # noqa: E402
from tools.lgbench import SHAPES, graph_time, operands  # noqa: E402


def main(masks):
    B, K, dx, dy = SHAPES["c4"]
    k = _kernels.get()
    device = torch.device("cuda:0")
    sets = [operands(B, K, dx, dy, torch.float32, device, seed=s) for s in range(4)]
    gen = torch.Generator().manual_seed(1)
    for o in sets:
        o["terms"] = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
        o["scales"] = (o["s_p"], o["s_g"], o["s_q"])
        o["lw"] = k.affine_logweight(o["x_prev"], o["x"], o["y"], *o["terms"], o["scales"])
        o["lse"] = k.logweight_lse(o["lw"], None, None, want_lw=False)[1]
        o["glse"] = torch.ones_like(o["lse"])
        lw = torch.randn(B, K, generator=gen, dtype=torch.float64).float().to(device)
        u = torch.rand(B, generator=gen, dtype=torch.float64).to(device)
        idx, _, _ = k.resample_step(lw, u, None, want_lse=False, want_child_end=True)
        o["idx"], o["ce"] = idx, idx._aesmc_child_end
    need = [True, False, False, True, False, True, False, True, True, False, False, False]
    state = {"i": 0}

    def fn():
        state["i"] = (state["i"] + 1) % len(sets)
        o = sets[state["i"]]
        return k.affine_step_backward(o["x_prev"], o["x"], o["y"], *o["terms"], o["scales"], need, o["lw"], o["lse"],
                                      grad_lse=o["glse"], ancestors=o["idx"], child_grad=o["eps"], child_end=o["ce"])

    for mask in masks:
        os.environ["AESMC_K14_PROBE"] = str(mask)
        print("probe {:3d}: {:7.1f} us".format(mask, graph_time(fn)), flush=True)


if __name__ == "__main__":
    main([int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 32, 64, 6, 14, 30, 31, 127])
