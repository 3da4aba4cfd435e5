"""hipGraph-timed launches of K14 (aesmc_affine_step_backward[_resampled]) at a BASELINE shape, one line per way the
gradient of x_t can arrive: nothing, as a summed tensor (`grad_x`), or per child with the children ranges (the
gather's backward folded in) — healthy and collapsed next-step ancestries.

    python tools/k14bench.py [--shape c4]      (AESMC_LG_CHILD_STAGE=0: lanes fetch their children's rows themselves)
"""
import argparse
import os
os.environ.setdefault("AESMC_MEASUREMENT_KNOBS", "1")      # the library reads AESMC_* knobs only beside this
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops  # noqa: E402
from tools.lgbench import SHAPES, graph_time, operands  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--shape", default="c4")
    parser.add_argument("--dims", default=None, help="B,K,d: another shape (rows of d values on both sides)")
    parser.add_argument("--only", default=None, help="run only the case whose label contains this")
    args = parser.parse_args()
    B, K, dx, dy = SHAPES[args.shape]
    if args.dims:
        B, K, dx = [int(v) for v in args.dims.split(",")]
        dy = dx
    k = _kernels.get()
    device = torch.device("cuda:0")
    sets = [operands(B, K, dx, dy, torch.float32, device, seed=s) for s in range(4)]
    gen = torch.Generator().manual_seed(1)
    for o in sets:
        terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
        scales = (o["s_p"], o["s_g"], o["s_q"])
        o["terms"], o["scales"] = terms, scales
        o["lw"] = k.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
        o["lse"] = k.logweight_lse(o["lw"], None, None, want_lw=False)[1]
        o["glse"] = torch.ones_like(o["lse"])
        for spread, tag in ((1.0, "healthy"), (6.0, "collapsed")):
            lw = (spread * torch.randn(B, K, generator=gen, dtype=torch.float64)).float().to(device)
            u = torch.rand(B, generator=gen, dtype=torch.float64).to(device)
            idx, _, _ = k.resample_step(lw, u, None, want_lse=False, want_child_end=True)
            o["idx_" + tag], o["ce_" + tag] = idx, idx._aesmc_child_end
    need = [True, False, False, True, False, True, False, True, True, False, False, False]
    state = {"i": 0}

    def launch(**kwargs):
        def fn():
            state["i"] = (state["i"] + 1) % len(sets)
            o = sets[state["i"]]
            extra = {key: (o[value] if isinstance(value, str) else value) for key, value in kwargs.items()}
            return k.affine_step_backward(o["x_prev"], o["x"], o["y"], *o["terms"], o["scales"], need, o["lw"], o["lse"],
                                          grad_lse=o["glse"], **extra)
        return fn

    esz, N = 4, B * K
    cases = [
        ("dense x_prev, nothing arrives at x_t", {}, 3 * dx + 1),
        ("dense x_prev, grad_x", {"grad_x": "eps"}, 4 * dx + 1),
        ("through ancestors, nothing arrives", {"ancestors": "idx_healthy"}, 3 * dx + 3),
        ("through ancestors, grad_x", {"ancestors": "idx_healthy", "grad_x": "eps"}, 4 * dx + 3),
        ("through ancestors, children (healthy)", {"ancestors": "idx_healthy", "child_grad": "eps", "child_end": "ce_healthy"},
         4 * dx + 4),
        ("through ancestors, children (collapsed)", {"ancestors": "idx_healthy", "child_grad": "eps",
                                                      "child_end": "ce_collapsed"}, 4 * dx + 4),
    ]
    print("B={} K={} d={}  child staging {}".format(B, K, dx, os.environ.get("AESMC_LG_CHILD_STAGE", "1")))
    for label, kwargs, words in cases:
        if args.only and args.only not in label:
            continue
        us = graph_time(launch(**kwargs))
        label = "{} [form {}]".format(label, k._lib.aesmc_test_last_step_backward_form())
        nbytes = esz * N * words
        print("{:58s} {:8.1f} us  {:7.1f} MB  {:5.2f} TB/s".format(label, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)
    # the launch the folding replaces
    gs = [torch.randn(B, K, dx, device=device) for _ in range(2)]
    us = 0.0 if args.only else graph_time(lambda: k.gather_backward(gs[0], sets[0]["idx_healthy"], sorted_index=True))
    if not args.only:
        print("{:48s} {:8.1f} us".format("segmented sum (gather's backward), healthy", us))


if __name__ == "__main__":
    main()
