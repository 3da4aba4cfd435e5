import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd
from aesmc_amd import _kernels, _ops, _philox
from tests.test_gpu_linear_gaussian import operands
from tests.test_gpu_round3 import _ancestors
dev = torch.device("cuda", 0)
k = _kernels.get(); type(k).DRAWN_MIN_PARTICLES = 0
for (B, K, dx, dy) in [(3, 700, 10, 10)]:
    n, o = operands(min(B, 8), min(K, 64), dx, dy, np.float32, dev, seed=3 * B + K + dx)
    gen = torch.Generator(device=dev).manual_seed(B + K)
    x_prev = torch.randn(B, K, dx, device=dev, generator=gen)
    y = torch.randn(B, dy, device=dev, generator=gen)
    off_q = torch.randn(B, dx, device=dev, generator=gen)
    idx = _ancestors(B, K, dev, seed=B + K, spread=1.0)
    off_p = torch.from_numpy(np.random.RandomState(6).randn(dx).astype(np.float32)).to(dev)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], off_q))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    torch.manual_seed(1000 + K)
    state = torch.cuda.get_rng_state(dev)
    eps = torch.empty(B, K, dx, device=dev).normal_()
    torch.cuda.set_rng_state(state, dev)
    res = _philox.reserve(B * K * dx, dev)
    moved = k.gather(x_prev, idx)
    want_x = torch.full_like(moved, float("nan"))
    want_lw = k.affine_propagate(moved, eps, y, *terms, scales, out_x=want_x)
    got_x = torch.full_like(moved, float("nan"))
    got_lw = k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=got_x, ancestors=idx)
    torch.cuda.synchronize()
    bad = (got_x != want_x) | torch.isnan(got_x)
    print("x mismatches", int(bad.sum()), "of", bad.numel(), "flags", k.read_flags(dev))
    rows = bad.any(dim=2).view(-1).nonzero().view(-1)
    print("bad particles", rows.numel(), rows[:40].tolist())
    if rows.numel():
        p = int(rows[0]); b, kk = divmod(p, K)
        print("first bad", p, "got", got_x[b, kk].tolist(), "want", want_x[b, kk].tolist())
        print("cols bad", bad.view(-1, dx)[rows].sum(dim=0).tolist())
    print("lw mismatches", int((got_lw != want_lw).sum()))
