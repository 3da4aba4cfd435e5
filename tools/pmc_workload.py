"""Workload for the PMC passes over the BENCH workload's own operands (rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE, one counter per run): the first timesteps of the same seeded model and data that
bench.py times, run through aesmc_amd.inference.infer, so every dispatch of the path's kernels in
the counter CSV worked on real log-weights and latents.  Before them, calibration launches with
exactly known traffic in the same access pattern: K3 with the identity index at the workload's
shape (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x on
gfx950 — the factor is measured, not assumed).

Usage: python tools/pmc_workload.py <workload> <proposal> [timesteps]
Dispatch order (resample_gather_kernel): 3 calibration launches.  Then `timesteps` steps of get_loss:
timesteps - 1 launches of ancestor_index_inv_kernel (K2; with the newest latent as its payload where the
model reads the resampled values) and of the propagation kernel (K16 / K15 / K9 + K10), then the backward:
timesteps - 1 launches of affine_step_backward_kernel (K14)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch

import aesmc_amd
from aesmc_amd import _kernels
import bench


def main(workload, proposal, timesteps=6):
    description, kind, dim, B, K, T, model_kwargs = bench.WORKLOADS[workload]
    device = torch.device("cuda", 0)
    k = _kernels.get()
    model = bench.build_model(kind, dim, device, aesmc_amd.state, proposal, os.environ.get("AESMC_CALLABLES", "affine"),
                              **model_kwargs)
    observations = model.simulate(T, B, seed=1)[:timesteps]
    value = torch.randn(B, K, dim, device=device)
    identity = torch.arange(K, device=device).unsqueeze(0).expand(B, K).contiguous()
    torch.cuda.synchronize()
    for _ in range(3):
        k.gather(value, identity)
    torch.cuda.synchronize()
    np.random.seed(0)
    torch.manual_seed(0)
    # as bench.py's step: a forward ELBO with the autograd graph recorded (the resampling launch then also writes the
    # children ranges the backward uses)
    loss = aesmc_amd.losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission,
                                     model.proposal)
    if os.environ.get("AESMC_PMC_BACKWARD", "1") != "0":
        loss.backward()      # K14 per timestep (the middle ones with the next step's children folded in)
    torch.cuda.synchronize()
    print("workload {} proposal {} B={} K={} d={}: loss {:.4f} over {} timesteps".format(
        workload, proposal, B, K, dim, float(loss), timesteps))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
