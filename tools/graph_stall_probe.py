"""Do hipGraph replays of PLAIN PyTorch launches over a large working set show the periodic stall the captured ELBO
shows above ~8 GB?  `steps` pairs of launches per graph, each writing its own `mb`-MB tensors allocated inside the
capture (as the ELBO's per-timestep tensors are); eight replays timed one by one.

    python tools/graph_stall_probe.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401  (the hipGraph setting the package runs under)


def main():
    dev = torch.device("cuda", 0)
    for steps, mb in ((100, 20), (100, 50), (100, 100), (200, 100)):
        src = torch.randn(mb * 2**20 // 4, device=dev)
        g = torch.cuda.CUDAGraph()
        keep = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                keep.append(src * 2.0)
        torch.cuda.current_stream().wait_stream(side)
        keep = []
        with torch.cuda.graph(g):
            x = src
            for _ in range(steps):
                y = x * 1.0001          # a fresh tensor per step, kept (as an autograd graph would keep it)
                z = torch.tanh(y)
                keep.append((y, z))
                x = z
        torch.cuda.synchronize()
        times = []
        for _ in range(8):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        print("%d steps x 2 tensors of %d MB (%.1f GB kept): replays of %s ms" % (
            steps, mb, steps * 2 * mb / 1024, " ".join("%.1f" % t for t in times)), flush=True)
        del g, keep, x, y, z
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
