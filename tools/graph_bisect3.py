import subprocess
import sys

CASE = r'''
import sys, torch, numpy as np
sys.path.insert(0, ".")
from aesmc_amd import graphs, losses, _kernels, state
from aesmc_amd.testing import models
dtype = torch.float64
model = models.LgssmNd(3, seed=0, dtype=dtype, validate_args=False).to("cuda")
obs = model.simulate(6, 8, seed=1)
parts = (model.initial, model.transition, model.emission, model.proposal)
variant = "{variant}"
k = _kernels.get()
x = torch.randn(8, 64, 3, device="cuda", dtype=dtype)
idx = torch.zeros(8, 64, dtype=torch.int64, device="cuda")
if variant == "nograd_del":
    with torch.no_grad():
        loss = losses.get_loss(obs, 64, "aesmc", *parts)
    del loss
elif variant == "gather_only":
    k.gather(x, idx); torch.cuda.synchronize()
elif variant == "k1_only":
    k.logweight_lse(x[..., 0].contiguous()); torch.cuda.synchronize()
elif variant == "k2_only":
    k.ancestor_index(x[..., 0].contiguous(), torch.rand(8, device="cuda", dtype=torch.float64)); torch.cuda.synchronize()
elif variant == "k4_only":
    k.normal_logprob_sum(x, x, x.abs() + 1); torch.cuda.synchronize()
elif variant == "flags_only":
    k.read_flags(torch.device("cuda", 0))
elif variant == "torch_only":
    y = (x @ torch.randn(3, 3, device="cuda", dtype=dtype)).sum().item()
elif variant == "side_stream_eager":
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        loss = losses.get_loss(obs, 64, "aesmc", *parts)
    torch.cuda.synchronize()
elif variant == "pinned_only":
    h = torch.empty(5, 8, dtype=torch.float64, pin_memory=True); d = torch.empty(5, 8, dtype=torch.float64, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize(); del h, d
g = graphs.GraphedLoss(obs, 64, "aesmc", *parts, backward={bwd})
print("captured", float(g()), float(g()))
'''
for variant, bwd in [("nograd_del", False), ("gather_only", False), ("k1_only", False), ("k2_only", False),
                     ("k4_only", False), ("flags_only", False), ("torch_only", False), ("side_stream_eager", False),
                     ("pinned_only", False)]:
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CASE.format(variant=variant, bwd=bwd)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    print(variant, bwd, "rc", r.returncode, tail, flush=True)
