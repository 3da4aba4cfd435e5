"""Workload for hardware-counter passes over the linear-Gaussian propagation kernels (rocprofv3 --pmc,
one counter set per run): K8 .. K12, K14, K15 at B=1024 K=4096 d=10,
3 launches each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aesmc_amd import _kernels
from lgbench import operands

k = _kernels.get()
dev = torch.device("cuda", 0)
shape = tuple(int(v) for v in os.environ.get("LG_SHAPE", "1024,4096,10,10").split(","))
o = operands(*shape, torch.float32, dev)
for _ in range(3):
    k.particle_affine(o["x_prev"], o["Q"], o["off_q"])
    k.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"])
    k.affine_logweight(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                       (o["s_p"], o["s_g"], o["s_q"]))
    k.particle_affine_backward(o["eps"], o["x_prev"], o["Q"])
    lw = k.affine_logweight(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                            (o["s_p"], o["s_g"], o["s_q"]))
    lse = k.logweight_lse(lw, None, None, want_lw=False)[1]
    need = [True, True, False, True, False, True, False, True, True, False, False, False]
    k.affine_logweight_backward(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                                (o["s_p"], o["s_g"], o["s_q"]), need, lw=lw, lse=lse, grad_lse=torch.ones_like(lse))
    k.affine_propagate(o["x_prev"], o["eps"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                       (o["s_p"], o["s_g"], o["s_q"]), out_x=torch.empty_like(o["x"]))
    need[1] = False
    k.affine_step_backward(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                           (o["s_p"], o["s_g"], o["s_q"]), need, lw, lse, grad_lse=torch.ones_like(lse), grad_x=o["eps"])
torch.cuda.synchronize()
print("done")
