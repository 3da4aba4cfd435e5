"""Which MEMSET nodes does a captured ELBO hold, and what do they zero?  Run under rocprofv3's kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/graph_memsets.py replay [B K T]
    python tools/graph_memsets.py summarize OUT/*/*_kernel_trace.csv

`replay` captures the north-star model's loss as GraphedLoss captures it — forward only, then forward + backward — and
replays each graph twice between marker kernels (a torch.zeros of a size nothing else uses).  A captured memset node
runs as the runtime's fill kernel (`__amd_rocclr_fillBufferAligned`); `summarize` lists, per graph, how many fills one
replay holds and which kernels follow them (what the zeroed buffer is for).
"""
import collections
import csv
import os
import sys


def replay(B, K, T):
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import aesmc_amd  # noqa: F401
    from aesmc_amd import graphs
    from aesmc_amd.testing.models import LgssmNd
    dev = torch.device("cuda", 0)
    marker = torch.rand(12347, device=dev)
    from aesmc_amd.testing.models import NonlinearSsm
    cases = [("linear-Gaussian, AffineNormal callables", lambda: LgssmNd(10, dtype=torch.float32, affine=True, validate_args=False).tune_proposal()),
             ("linear-Gaussian, Normal(x @ W.t() + c, s) callables", lambda: LgssmNd(10, dtype=torch.float32, affine=False, validate_args=False).tune_proposal()),
             ("nonlinear SSM with an MLP proposal (configs[3])", lambda: NonlinearSsm(10, dtype=torch.float32, validate_args=False))]
    for label, make in cases:
      for backward in (False, True):
        print("CASE {} | backward={}".format(label, backward), flush=True)
        model = make().to(dev)
        for p in model.parameters():
            p.requires_grad_(backward)
        observations = model.simulate(T, B, seed=1)
        np.random.seed(0)
        torch.manual_seed(0)
        graphed = graphs.GraphedLoss(observations, K, "aesmc", model.initial, model.transition, model.emission,
                                     model.proposal, backward=backward, verify_replays=0)
        torch.cuda.synchronize()
        for _ in range(2):
            marker.erfinv_()      # a marker launch in front of every replay (a kernel nothing else here uses)
            graphed(observations)
            torch.cuda.synchronize()
        marker.erfinv_()
        marker.erfinv_()          # two in a row: the end of a case
        torch.cuda.synchronize()
        del graphed


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # marker launches: elementwise fills of PyTorch's own (FillFunctor) — the replays lie between consecutive ones
    marks = [i for i, n in enumerate(names) if "erfinv" in n]
    print("{} dispatches, {} marker launches".format(len(names), len(marks)))
    seen = case = 0
    for a, b in zip(marks, marks[1:]):
        span = names[a + 1:b]
        if not span:
            case += 1
            seen = 0
            continue
        if seen == 0:
            print("case {} (in the order tools/graph_memsets.py replay printed them)".format(case))
        seen += 1
        fills = [i for i, n in enumerate(span) if "fillBuffer" in n]
        follows = collections.Counter()
        for i in fills:
            nxt = [n for n in span[i + 1:i + 3] if "fillBuffer" not in n]
            follows[(nxt[0] if nxt else "<end>")[:110]] += 1
        print("replay {}: {} dispatches, {} runtime fill kernels (memset nodes)".format(seen, len(span), len(fills)))
        for name, count in follows.most_common(12):
            print("    {:4d} x followed by {}".format(count, name))


if __name__ == "__main__":
    if sys.argv[1] == "replay":
        replay(*([int(v) for v in sys.argv[2:]] or [8, 256, 6]))
    else:
        summarize(sys.argv[2])
