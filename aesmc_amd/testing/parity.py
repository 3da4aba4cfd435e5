"""End-to-end agreement of a float32 `infer` run with a run of the reference itself, measured two ways.

The reference builds the resampler's CDF in the input dtype (float32 NumPy / SciPy: aesmc/inference.py:253-264),
this package in float64: on float32 log-weights a comparison that sits within rounding
noise of flipping may come out differently, and from the first such flip on the two runs are different — equally
valid — particle systems, so "compare everything at the end" says little.  Hence:

  * FREE-RUNNING: the device run on the reference's inputs and replayed draws, compared as is — per-step fraction
    of equal ancestor indices and |delta log Z|: what a user sees;
  * TEACHER-FORCED: the same run with the reference's ancestor indices substituted after every resampling launch
    (the launch's own indices are kept for counting), so the particle systems stay aligned for the whole sequence
    and EVERY step's log-weights can be held against the reference's to float32 rounding — a flip no longer
    switches the comparison off — while the flips themselves are counted per step.

Used by tests/test_gpu_noise_and_lazy_latents.py (bounds) and bench.py (`extras.fp32_fixture_parity`: the achieved numbers).
No oracle import: the fixture arrays are handed in.
"""
import numpy as np
import torch

from .. import _ops, inference
from . import replay


def _run(parts, observations, num_particles, tape, forced_indices=None):
    own = []
    real = _ops.resample_step

    def forcing(log_weight, uniforms, payload=None, want_lse=False, pending=None, **unused):
        index, lse, moved = real(log_weight, uniforms, None, want_lse=want_lse, pending=pending)
        own.append(index)
        step = len(own) - 1
        forced = torch.from_numpy(np.ascontiguousarray(forced_indices[step])).to(index.device)
        forced._aesmc_sorted = True
        return forced, lse, None

    if forced_indices is not None:
        _ops.resample_step = forcing
    try:
        with replay.replay(tape), torch.no_grad():
            result = inference.infer("smc", observations, parts["initial"], parts["transition"], parts["emission"],
                                     parts["proposal"], num_particles, return_log_marginal_likelihood=True,
                                     return_latents=False, return_log_weights=True, return_ancestral_indices=True)
    finally:
        _ops.resample_step = real
    return result, own


def float32_fixture_parity(parts, observations, num_particles, tape, reference):
    """`reference`: dict with the reference run's `log_weights` [T] arrays, `indices` [T-1] arrays and `lml` [B].
    Returns the achieved numbers (plain Python floats / lists)."""
    steps = len(reference["indices"])
    free, _ = _run(parts, observations, num_particles, tape)
    agreement = [float((free["ancestral_indices"][t].cpu().numpy() == reference["indices"][t]).mean())
                 for t in range(steps)]
    lml = free["log_marginal_likelihood"].double().cpu().numpy()
    want_lml = np.asarray(reference["lml"], dtype=np.float64)
    forced, own = _run(parts, observations, num_particles, tape, forced_indices=reference["indices"])
    worst_lw, flips = [], []
    for t, want in enumerate(reference["log_weights"]):
        got = forced["log_weights"][t].double().cpu().numpy()
        want = np.asarray(want, dtype=np.float64)
        worst_lw.append(float(np.max(np.abs(got - want) / (1.0 + np.abs(want)))))
    for t in range(steps):
        flips.append(int((own[t].cpu().numpy() != reference["indices"][t]).sum()))
    forced_lml = forced["log_marginal_likelihood"].double().cpu().numpy()
    return {
        "free_running_index_agreement_per_step": agreement,
        "free_running_index_agreement_min": min(agreement) if agreement else 1.0,
        "free_running_index_agreement_mean": float(np.mean(agreement)) if agreement else 1.0,
        "free_running_first_flip_step": next((t for t, a in enumerate(agreement) if a < 1.0), None),
        "free_running_rel_dlogZ": float(np.max(np.abs(lml - want_lml) / (1.0 + np.abs(want_lml)))),
        "teacher_forced_max_rel_dlogw_per_step": worst_lw,
        "teacher_forced_max_rel_dlogw": max(worst_lw),
        "teacher_forced_flips_per_step": flips,
        "teacher_forced_flip_rate": sum(flips) / float(max(1, steps * reference["indices"][0].size)) if steps else 0.0,
        "teacher_forced_rel_dlogZ": float(np.max(np.abs(forced_lml - want_lml) / (1.0 + np.abs(want_lml)))),
    }
