"""Differentiable training objective with the interface of the reference's aesmc/losses.py."""
import torch

from . import inference

_ALGORITHM_TO_INFERENCE = {"iwae": "is", "aesmc": "smc"}


def get_loss(observations, num_particles, algorithm, initial, transition,
             emission, proposal):
    """Negative batch mean of the log-marginal-likelihood estimate: algorithm 'iwae' uses
    importance sampling, 'aesmc' uses SMC (aesmc/losses.py:5-65).  The callables follow the
    contract documented at `aesmc_amd.inference.infer`.  Call `.backward()` on the result."""
    if algorithm not in _ALGORITHM_TO_INFERENCE:
        # the reference falls through to an unbound local here (losses.py:45-50)
        raise UnboundLocalError("algorithm must be iwae or aesmc. currently = {}".format(algorithm))
    result = inference.infer(
        inference_algorithm=_ALGORITHM_TO_INFERENCE[algorithm], observations=observations,
        initial=initial, transition=transition, emission=emission, proposal=proposal,
        num_particles=num_particles, return_log_marginal_likelihood=True, return_latents=False,
        return_original_latents=False, return_log_weight=False, return_log_weights=False,
        return_ancestral_indices=False)
    return -torch.mean(result["log_marginal_likelihood"])
