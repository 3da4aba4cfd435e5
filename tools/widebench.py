"""K17 + K18 (the wide linear-Gaussian step: rows of 128 values on the fp32 matrix cores) at configs[4]'s shape, beside
the launches it replaces there (K3 gather, three library GEMMs with their offsets' adds, the draw from given noise, the
three-Normal log-weight), hipGraph-timed.

    python tools/widebench.py [B K]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _philox  # noqa: E402
from tools.lgbench import graph_time  # noqa: E402


def main(B, K):
    d = 128
    k = _kernels.get()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, K, d, device=dev, generator=g)
    eps = torch.randn(B, K, d, device=dev, generator=g)
    y = torch.randn(B, d, device=dev, generator=g)
    A = 0.9 * torch.eye(d, device=dev) + 0.05 * torch.randn(d, d, device=dev, generator=g)
    Q = 0.45 * torch.eye(d, device=dev) + 0.05 * torch.randn(d, d, device=dev, generator=g)
    C = 0.1 * torch.randn(d, d, device=dev, generator=g)
    off_q, off_g = torch.randn(B, d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
    scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
    lw = torch.randn(B, K, device=dev, generator=g)
    u = torch.rand(B, device=dev, dtype=torch.float64, generator=g)
    anc = k.ancestor_index(lw, u)
    out_x = torch.empty_like(x)
    terms = ((A, None), (C, off_g), (Q, off_q))
    t_wide = graph_time(lambda: k.affine_propagate_wide(x, eps, y, *terms, scales, out_x, ancestors=anc))
    assert k.affine_propagate_wide(x, eps, y, *terms, scales, out_x, ancestors=anc) is not None
    res = _philox.reserve(B * K * d, dev)
    t_drawn = None
    if k.affine_propagate_wide(x, res, y, *terms, scales, out_x, ancestors=anc) is not None:
        t_drawn = graph_time(lambda: k.affine_propagate_wide(x, res, y, *terms, scales, out_x, ancestors=anc))

    def pieces():
        moved = k.gather(x, anc)
        loc_q = moved @ Q.t() + off_q.unsqueeze(1)
        xt = loc_q + eps * scales[2]
        loc_p = moved @ A.t()
        loc_g = xt @ C.t() + off_g
        return xt, loc_p, loc_g
    t_pieces = graph_time(pieces)
    nbytes = 4 * B * K * (3 * d + 1) + 8 * B * K
    flops = 3 * 2.0 * B * K * d * d
    print("B={} K={} d={}: K17 + K18 {:.1f} us = {:.1f} TFLOP/s fp32 on the matrix cores, {:.2f} TB/s of {:.0f} MB; gather + three "
          "GEMMs + adds + draw (no log-weight kernel) {:.1f} us".format(B, K, d, t_wide, flops / t_wide / 1e6,
                                                                       nbytes / t_wide / 1e6, nbytes / 1e6, t_pieces), flush=True)
    if t_drawn is not None:
        print("   with the noise formed in the launch: {:.1f} us = {:.1f} TFLOP/s".format(t_drawn, flops / t_drawn / 1e6), flush=True)


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]] or [64, 16384]
    main(*args)
