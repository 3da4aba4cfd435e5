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

    def build_parts(self, state, device):
        """This package's counterpart model, loaded with the fixture's parameters."""
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
        elif meta["model"] == "lgssm_nd":
            model = models.LgssmNd(meta["dim"], proposal_scale=meta["proposal_scale"],
                                   seed=meta["seed"], dtype=dtype, state=state)
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
                   "resampler_k1_f64", "resampler_degenerate_f64"]
