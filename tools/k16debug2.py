import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd
from aesmc_amd import _kernels, _ops, _philox
from tests.test_gpu_round3 import _ancestors
dev = torch.device("cuda", 0)
k = _kernels.get(); type(k).DRAWN_MIN_PARTICLES = 0
B, K, dx, dy = 3, 700, 10, 10
x_prev = (torch.arange(B * K, device=dev, dtype=torch.float32).view(B, K, 1) * 16 + torch.arange(dx, device=dev, dtype=torch.float32)).contiguous()
y = torch.zeros(B, dy, device=dev)
idx = _ancestors(B, K, dev, seed=B + K, spread=1.0)
I = torch.eye(dx, device=dev)
terms = ((I, None), (I, None), (I, None))
scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 1.0, 1e-30))
res = _philox.reserve(B * K * dx, dev)
got_x = torch.full_like(x_prev, float("nan"))
got_lw = k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=got_x, ancestors=idx)
torch.cuda.synchronize()
want = k.gather(x_prev, idx)
bad = got_x != want
print("mismatches", int(bad.sum()), "of", bad.numel())
g = got_x.view(-1, dx).cpu(); wv = want.view(-1, dx).cpu()
for p in list(range(0, 6)) + [63, 64, 65, 127, 128, 129, 700, 701]:
    print(p, "anc", int(idx.view(-1)[p]), "got", [int(v) if v == v else None for v in g[p].tolist()], "want row", int(wv[p, 0]) // 16)
