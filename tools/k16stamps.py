"""Where an item's cycles go inside the fused propagation launch: a build with -DAESMC_K16_PROBES stamps s_memtime
at the phase boundaries of one particle wavefront and one noise wavefront per workgroup and leaves the sums in a
buffer whose address it is given through AESMC_K16_STAMPS.  Shares, not durations: the stamps' own waits forbid
overlaps the product build has.
NOTE (round 6): the probe / stamp code this script drives was removed from the product translation units (VERDICT r05,
hygiene).  The instrumented kernels are the tree at commit 549e68a: to repeat the experiment, check that commit's
aesmc_amd/csrc/ out into tools/exp/ (git-ignored), build it with AESMC_HIPCC_FLAGS=-DAESMC_K16_PROBES (or -DAESMC_K14_PROBES)
and AESMC_PROBE_BUILD=1, and point the loader at that library.  The results it produced are under profiles/ (r04_k16_stamps.txt,
r04_k14_probes.txt, r05_pmc_k14_c4.txt, ...).
"""
import os
os.environ.setdefault("AESMC_MEASUREMENT_KNOBS", "1")      # the library reads AESMC_* knobs only beside this
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402,F401
from aesmc_amd import _kernels, _ops, _philox  # noqa: E402

B, K, d = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 4096, 10))]
dev = torch.device("cuda", 0)
k = _kernels.get()
type(k).DRAWN_MIN_PARTICLES = 0
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)
x_prev, out_x, lw = r(B, K, d), torch.empty(B, K, d, device=dev), r(B, K)
u = torch.rand(B, device=dev, dtype=torch.float64, generator=gen)
y = r(B, d)
eye = torch.eye(d, device=dev)
A, C, Q = 0.9 * eye + 0.01 * r(d, d), eye + 0.01 * r(d, d), 0.45 * eye + 0.01 * r(d, d)
terms = ((A, None), (C, None), (Q, r(B, d)))
scales = tuple(torch.tensor(v, device=dev) for v in (1.0, 0.5, 0.7))
idx = _ops.ancestor_index(lw, u)
res = _philox.reserve(B * K * d, dev)
stamps = torch.zeros(512 * 2 * 16, dtype=torch.int64, device=dev)
P_NAMES = ["prologue barrier", "q/p matrix products + scratch stores", "row + ancestor loads sent", "rows: draw, residuals, chains",
           "emission products", "emission chain + log-weight", "x_t copy-out", "wait at the barrier", "next window"]
N_NAMES = ["loop head", "table loads sent", "draws + table store", "wait at the barrier"]
for mask in [int(v) for v in sys.argv[4:]] or [0]:
    os.environ["AESMC_K16_PROBE"] = str(mask)
    os.environ.pop("AESMC_K16_STAMPS", None)
    for _ in range(3):
        k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=out_x, ancestors=idx)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=out_x, ancestors=idx)
    b.record()
    torch.cuda.synchronize()
    plain = a.elapsed_time(b) * 1e3
    os.environ["AESMC_K16_STAMPS"] = str(stamps.data_ptr())
    stamps.zero_()
    a.record()
    k.affine_propagate_drawn(x_prev, res, y, *terms, scales, out_x=out_x, ancestors=idx)
    b.record()
    torch.cuda.synchronize()
    table = stamps.view(512, 2, 16).double().cpu()
    print("probe {}: {:.1f} us unstamped, {:.1f} us stamped; cycles per workgroup (median over workgroups), share".format(
        mask, plain, a.elapsed_time(b) * 1e3))
    for role, names in ((0, P_NAMES), (1, N_NAMES)):
        med = table[:, role, :len(names)].median(dim=0).values
        total = float(med.sum())
        cyc, real = float(table[:, role, 14].median()), float(table[:, role, 15].median())
        print("  {} wavefront: {:.0f} cycles in all; shader clock {:.2f} GHz ({:.0f} cycles in {:.1f} us of the 100 MHz counter)".format(
            "particle" if role == 0 else "noise", total, cyc / max(real, 1) * 0.1, cyc, real / 100))
        for name, value in zip(names, med.tolist()):
            print("    {:42s} {:10.0f}  {:5.1f} %".format(name, value, 100 * value / max(total, 1)))
