"""Does torch's normal_ (VALU-bound Philox + Box-Muller) overlap with the path's kernels when it runs on a
side stream?  Times [normal_ ; K9 ; K10 ; fused step] back to back on one stream against the same with
normal_ (for the NEXT step) on a second stream, at B=1024 K=4096 d=10."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesmc_amd import _kernels
from lgbench import operands

k = _kernels.get()
dev = torch.device("cuda", 0)
o = operands(1024, 4096, 10, 10, torch.float32, dev)
lw = torch.randn(1024, 4096, device=dev)
u = torch.rand(1024, device=dev, dtype=torch.float64)
eps_a, eps_b = torch.empty_like(o["eps"]), torch.empty_like(o["eps"])
terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
scales = (o["s_p"], o["s_g"], o["s_q"])
side = torch.cuda.Stream()


def step_serial(eps):
    eps.normal_()
    x = k.affine_rsample(o["x_prev"], o["Q"], o["off_q"], eps, o["s_q"])
    k.affine_logweight(o["x_prev"], x, o["y"], *terms, scales)
    k.resample_step(lw, u, x, want_lse=True)


def step_overlapped(eps_now, eps_next, ready):
    main = torch.cuda.current_stream()
    main.wait_event(ready)                       # eps_now was drawn on the side stream during the previous step
    x = k.affine_rsample(o["x_prev"], o["Q"], o["off_q"], eps_now, o["s_q"])
    side.wait_stream(main)                       # the next draw may start once K9 is in flight ... after it, strictly
    with torch.cuda.stream(side):
        eps_next.normal_()
        done = torch.cuda.Event()
        done.record(side)
    k.affine_logweight(o["x_prev"], x, o["y"], *terms, scales)
    k.resample_step(lw, u, x, want_lse=True)
    return done


for _ in range(5):
    step_serial(eps_a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    step_serial(eps_a if i % 2 == 0 else eps_b)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / 200
ready = torch.cuda.Event()
eps_a.normal_()
ready.record()
for i in range(5):
    ready = step_overlapped(eps_a if i % 2 == 0 else eps_b, eps_b if i % 2 == 0 else eps_a, ready)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    ready = step_overlapped(eps_a if i % 2 == 0 else eps_b, eps_b if i % 2 == 0 else eps_a, ready)
torch.cuda.synchronize()
overlapped = (time.perf_counter() - t0) / 200
print("per step: serial {:.1f} us, noise on a side stream {:.1f} us".format(1e6 * serial, 1e6 * overlapped))
