"""Loads tests/golden/*.npz fixtures (captured from the reference by oracle/capture_golden.py)."""
import json
import os

import numpy as np
import torch

from aesmc_amd.testing import models
from aesmc_amd.testing.replay import Tape

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.arrays = {k: data[k] for k in data.files}
        self.meta = json.loads(str(self.arrays.pop("meta")))
        self.name = name

    def __getitem__(self, key):
        return self.arrays[key]

    def series(self, prefix):
        out, i = [], 0
        while "{}_{}".format(prefix, i) in self.arrays:
            out.append(self.arrays["{}_{}".format(prefix, i)])
            i += 1
        return out

    @property
    def dtype(self):
        return getattr(torch, self.meta["dtype"])

    def tape(self):
        return Tape(self.series("normal"), self.series("uniform"))

    def observations(self, device):
        return [torch.from_numpy(o).to(device) for o in self.series("obs")]

    def build_parts(self, state, device, affine=False, frozen=False):
        """This package's counterpart model, loaded with the fixture's parameters (`affine`: the d-dimensional
        LGSSM with its callables returning AffineNormal — the same model, stated for kernels K9 / K10 / K12;
        `frozen`: the very callables the fixture was generated with, oracle/fixture_models.py — what the CPU
        port must reproduce to the last place)."""
        meta, dtype = self.meta, self.dtype
        if meta["model"] == "lgssm1d":
            parts = {
                "initial": models.Lgssm1dInitial(*meta["initial"]),
                "transition": models.Lgssm1dTransition(0.0, meta["transition_scale"], state=state),
                "emission": models.Lgssm1dEmission(0.0, meta["emission_scale"], state=state),
                "proposal": models.Lgssm1dProposal(*meta["proposal_scales"], state=state),
            }
        elif meta["model"] == "gaussian":
            parts = {"initial": models.GaussianPrior(0.0, meta["prior_std"]), "transition": None,
                     "emission": models.GaussianLikelihood(1.0),
                     "proposal": models.GaussianInferenceNetwork(0.0, 0.0, 1.0)}
        elif meta["model"] == "lgssm_nd" and frozen:
            from oracle import fixture_models
            model = fixture_models.FixtureLgssmNd(meta["dim"], state, proposal_scale=meta["proposal_scale"],
                                                  seed=meta["seed"], dtype=dtype)
            parts = {"initial": model.initial, "transition": model.transition,
                     "emission": model.emission, "proposal": model.proposal, "model": model}
        elif meta["model"] == "lgssm_nd":
            model = models.LgssmNd(meta["dim"], proposal_scale=meta["proposal_scale"],
                                   seed=meta["seed"], dtype=dtype, state=state, affine=affine)
            parts = {"initial": model.initial, "transition": model.transition,
                     "emission": model.emission, "proposal": model.proposal, "model": model}
        else:
            raise KeyError(meta["model"])
        named = {}
        for part_name, part in parts.items():
            if isinstance(part, torch.nn.Module):
                part.to(device=device, dtype=dtype)
                for pname, p in part.named_parameters():
                    named["{}.{}".format(part_name, pname)] = p
        assert sorted(named) == meta["param_names"], (sorted(named), meta["param_names"])
        with torch.no_grad():
            for pname, p in named.items():
                p.copy_(torch.from_numpy(self["param_" + pname]).to(device=device, dtype=dtype))
        return parts, named


INFER_CASES = ["c1_lgssm1d_smc_f32", "c1_lgssm1d_smc_f64", "c1_lgssm1d_smc_stock_f32",
               "c1_lgssm1d_is_f32", "lgssm3d_smc_f32", "lgssm3d_smc_f64", "lgssm3d_is_f32",
               "lgssm10d_smc_f64", "gaussian_iwae_f32"]
RESAMPLER_CASES = ["resampler_k1000_s1_f64", "resampler_k1000_s5_f64", "resampler_k4096_f64",
                   "resampler_k1000_s1_f32", "resampler_k4096_f32", "resampler_edge_f64",
                   "resampler_k1_f64", "resampler_degenerate_f64", "resampler_k16384_f64",
                   "resampler_k16384_f32"]
# large fixtures that keep inputs, per-step log-weights and indices but not the latents
LIGHT_INFER_CASES = ["lgssm10d_k1024_smc_f32", "lgssm10d_k4096_smc_f32"]
TRAIN_CASES = ["train_iwae_gaussian", "train_aesmc_lgssm1d"]


def float32_flip_bound(num_particles):
    """Largest fraction of float32 ancestor indices allowed to differ (by one) from the reference's:
    twice the rate SURVEY.md section 7 (hard part 1) measured for ANY float64-CDF implementation
    against the reference's float32 NumPy / SciPy pipeline — 31, 311, 2191 of 262144 indices at
    K = 1024, 4096, 16384 — interpolated as a power law between those particle counts."""
    import math
    table = [(1024, 31 / 262144.0), (4096, 311 / 262144.0), (16384, 2191 / 262144.0)]
    if num_particles <= table[0][0]:
        return 2.0 * table[0][1]
    for (k0, r0), (k1, r1) in zip(table, table[1:]):
        if num_particles <= k1:
            t = math.log(num_particles / k0) / math.log(k1 / k0)
            return 2.0 * math.exp(math.log(r0) + t * (math.log(r1) - math.log(r0)))
    (k0, r0), (k1, r1) = table[-2:]
    slope = math.log(r1 / r0) / math.log(k1 / k0)
    return min(1.0, 2.0 * r1 * (num_particles / k1) ** slope)


def mismatch_margin(log_weight, uniforms, idx_a, idx_b):
    """How far from flipping the comparisons are on which two sets of ancestor indices disagree:
    the largest |c[j] - pos[k]| over every CDF entry j that lies between the two answers for
    position k (float64 CDF of `log_weight`, positions (u + k) / K).  Systematic resampling counts
    the CDF entries below each position; the reference builds that CDF in the input dtype, so in
    float32 its entries carry ~1e-6 of rounding noise and only comparisons closer than that can come
    out differently — by one index, or by several where a stretch of particles has weight below the
    noise.  0.0 when the two sets are equal."""
    log_weight = np.asarray(log_weight, dtype=np.float64)
    K = log_weight.shape[1]
    w = np.exp(log_weight - log_weight.max(axis=1, keepdims=True))
    c = np.cumsum(w, axis=1)
    c /= c[:, -1:]
    pos = (np.asarray(uniforms, dtype=np.float64).reshape(-1, 1) + np.arange(K)) / K
    worst = 0.0
    for b, k in np.argwhere(np.asarray(idx_a) != np.asarray(idx_b)):
        lo, hi = sorted((int(idx_a[b, k]), int(idx_b[b, k])))
        worst = max(worst, float(np.abs(c[b, lo:hi] - pos[b, k]).max()))
    return worst


FLOAT32_CDF_NOISE = 1e-5   # bound on `mismatch_margin` for float32 log-weights (K <= 16384)
