"""Does a hipGraph replay of the north-star ELBO slow down with the SIZE OF THE CAPTURE'S MEMORY POOL?  The same
forward ELBO captured twice per batch size: with the autograd graph recorded (every timestep's tensors stay in the
pool) and with parameters that need no gradient (temporaries are reused).  Prints the time of each of eight replays (synchronised one by one) and the pool size.

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
    import gc
    if os.environ.get("PROBE_NOGC"):
        gc.disable()
    marks = {}

    def on_gc(phase, info):      # every collection of the cyclic collector, with its duration
        if phase == "start":
            marks["t"] = time.perf_counter()
        else:
            print("      [gc generation %d: %.1f ms]" % (info["generation"], (time.perf_counter() - marks["t"]) * 1e3),
                  flush=True)
    gc.callbacks.append(on_gc)
    if os.environ.get("PROBE_MALLOPT"):      # keep freed host memory in the heap: no munmap / trim while the GPU runs
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        print("mallopt", libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30), flush=True)      # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD
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
            each = []
            for _ in range(8):
                t0 = time.perf_counter()
                graphed()
                torch.cuda.synchronize()
                each.append((time.perf_counter() - t0) * 1e3)
            print("B=%d autograd graph %s: replays of %s ms, pool %.1f GB" % (
                B, "recorded" if grad else "none", " ".join("%.1f" % ms for ms in each),
                (torch.cuda.memory_reserved() - before) / 2**30), flush=True)
            # where a slow replay spends its time: the host's part of the call (uniforms, generator state, the launch
            # of the graph) against the wait for the device afterwards
            parts = []
            for _ in range(8):
                t0 = time.perf_counter()
                graphed._refill()
                if graphed.noise is not None:
                    graphed.noise.upload()
                t1 = time.perf_counter()
                graphed.graph.replay()
                t2 = time.perf_counter()
                if graphed.noise is not None:
                    graphed.noise.advance()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                parts.append("%.1f+%.1f+%.1f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            print("      inputs + graph launch + wait: %s" % " ".join(parts), flush=True)
            feed = graphed.feed
            if feed is not None:      # the refill's own statements, one by one (ms each; a replay + sync after every refill)
                rows = []
                for _ in range(9):
                    slot = feed.turn
                    feed.turn ^= 1
                    t = [time.perf_counter()]
                    if feed.uploaded[slot] is not None:
                        feed.uploaded[slot].synchronize()
                    t.append(time.perf_counter())
                    block = np.random.uniform(size=[feed.num_draws, feed.batch_size, 1])[:, :, 0]
                    t.append(time.perf_counter())
                    feed.host[slot].copy_(torch.from_numpy(np.ascontiguousarray(block)))
                    t.append(time.perf_counter())
                    feed.dev.copy_(feed.mapped[slot] if feed.mapped[slot] is not None else feed.host[slot], non_blocking=True)
                    event = torch.cuda.Event()
                    event.record(torch.cuda.current_stream(feed.dev.device))
                    feed.uploaded[slot] = event
                    t.append(time.perf_counter())
                    torch.cuda.synchronize()
                    t.append(time.perf_counter())
                    graphed.graph.replay()
                    torch.cuda.synchronize()
                    t.append(time.perf_counter())
                    rows.append("/".join("%.1f" % ((b - a) * 1e3) for a, b in zip(t[:-1], t[1:])))
                print("      event wait / numpy draw / write pinned / enqueue copy / sync / replay: %s" % "  ".join(rows),
                      flush=True)
            if feed is not None:      # the same refill without any large host allocation: row by row into the pinned block
                views = [block.numpy() for block in feed.host]
                times = []
                for _ in range(12):
                    slot = feed.turn
                    feed.turn ^= 1
                    if feed.uploaded[slot] is not None:
                        feed.uploaded[slot].synchronize()
                    for row in range(feed.num_draws):
                        views[slot][row] = np.random.uniform(size=[feed.batch_size, 1])[:, 0]
                    feed.dev.copy_(feed.mapped[slot] if feed.mapped[slot] is not None else feed.host[slot], non_blocking=True)
                    event = torch.cuda.Event()
                    event.record(torch.cuda.current_stream(feed.dev.device))
                    feed.uploaded[slot] = event
                    t0 = time.perf_counter()
                    graphed.graph.replay()
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                print("      refill row by row (no large host allocation), then replay: %s" % " ".join("%.1f" % t for t in times),
                      flush=True)
            pairs = []      # new uniforms, then the same replay twice: is a slow replay slow again on the same numbers?
            for _ in range(8):
                graphed._refill()
                torch.cuda.synchronize()
                two = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    graphed.graph.replay()
                    torch.cuda.synchronize()
                    two.append((time.perf_counter() - t0) * 1e3)
                pairs.append("%.1f,%.1f" % tuple(two))
            print("      first and second replay on the same uniforms: %s" % "  ".join(pairs), flush=True)
            for idle_ms in (0.5, 2.0, 10.0):      # nothing but an idle device between replays of the same numbers
                times = []
                for _ in range(9):
                    time.sleep(idle_ms / 1e3)
                    t0 = time.perf_counter()
                    graphed.graph.replay()
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                print("      graph only after %.1f ms of idle device: %s" % (idle_ms, " ".join("%.1f" % t for t in times)),
                      flush=True)
            for what in ("graph only", "graph + uniforms", "graph + generator state"):
                times = []
                for _ in range(9):
                    t0 = time.perf_counter()
                    if what == "graph + uniforms":
                        graphed._refill()
                    if what == "graph + generator state" and graphed.noise is not None:
                        graphed.noise.upload()
                    graphed.graph.replay()
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                print("      %s: %s" % (what, " ".join("%.1f" % t for t in times)), flush=True)
            del graphed, model, observations
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [320, 384, 512])
