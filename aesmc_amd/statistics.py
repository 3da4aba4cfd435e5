"""Particle diagnostics and prior sampling with the interface of the reference's
aesmc/statistics.py.  Off the ELBO hot path.  Values on the HIP device come from kernel K7 (one
pass over the particles: ESS, weighted mean and second moment); when gradients are wanted the same
quantities are PyTorch expressions over K1's softmax, as differentiable as the reference's."""
import torch

from . import _kernels
from . import _ops
from . import math
from . import state


def _broadcast_weights(weights, like):
    """[B, K] -> [B, K, 1, ..., 1] so it multiplies a [B, K, ...] tensor."""
    return weights.reshape(weights.shape + (1,) * (like.dim() - 2))


def empirical_expectation(value, log_weight, f):
    """sum_k w_k f(value[:, k]) with w = softmax(log_weight, dim=1); `f` maps
    [batch_size, ...] -> [batch_size, ...] (aesmc/statistics.py:7-44)."""
    assert value.size()[:2] == log_weight.size()
    weights = math.exponentiate_and_normalize(log_weight, dim=1)
    total = None
    for particle in range(weights.size(1)):
        term = f(value[:, particle])
        w = weights[:, particle].reshape((-1,) + (1,) * (term.dim() - 1))
        total = w * term if total is None else total + w * term
    return total


def _summary_kernel_applies(log_weight, value=None):
    """Kernel K7 computes values only; whoever needs gradients keeps the PyTorch expression."""
    tensors = [log_weight] if value is None else [log_weight, value]
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return False
    if not (log_weight.is_cuda and log_weight.dtype in (torch.float32, torch.float64) and
            log_weight.dim() == 2 and log_weight.size(1) > 0):
        return False
    return value is None or (torch.is_tensor(value) and value.is_cuda and value.dtype == log_weight.dtype
                             and value.device == log_weight.device)


def empirical_mean(value, log_weight):
    """Weighted particle mean over dim 1 (aesmc/statistics.py:47-60): one pass of kernel K7 for
    values on the HIP device, the differentiable PyTorch reduction when gradients are wanted."""
    assert value.size()[:2] == log_weight.size()
    if _summary_kernel_applies(log_weight, value):
        return _kernels.get().particle_summary(log_weight.detach(), value.detach(), want_mean=True)[1]
    weights = math.exponentiate_and_normalize(log_weight, dim=1)
    return torch.sum(_broadcast_weights(weights, value) * value, dim=1)


def empirical_variance(value, log_weight):
    """Weighted particle variance E[x^2] - E[x]^2 (aesmc/statistics.py:63-76)."""
    assert value.size()[:2] == log_weight.size()
    if _summary_kernel_applies(log_weight, value):
        _, mean, second = _kernels.get().particle_summary(log_weight.detach(), value.detach(),
                                                          want_mean=True, want_second=True)
        return second - mean ** 2
    weights = _broadcast_weights(math.exponentiate_and_normalize(log_weight, dim=1), value)
    mean = torch.sum(weights * value, dim=1)
    return torch.sum(weights * value ** 2, dim=1) - mean ** 2


def log_ess(log_weight):
    """log effective sample size, 2 lse(lw) - lse(2 lw), over particles
    (aesmc/statistics.py:79-91); accepts [batch_size, num_particles] or [num_particles]."""
    rows = log_weight if log_weight.dim() == 2 else log_weight.unsqueeze(0)
    if _summary_kernel_applies(rows):
        value = _kernels.get().particle_summary(rows.detach(), want_log_ess=True)[0]
        return value if log_weight.dim() == 2 else value.squeeze(0)
    value = 2 * _ops.row_logsumexp(rows) - _ops.row_logsumexp(2 * rows)
    return value if log_weight.dim() == 2 else value.squeeze(0)


def ess(log_weight):
    """Effective sample size (aesmc/statistics.py:94-104)."""
    return torch.exp(log_ess(log_weight))


def sample_from_prior(initial, transition, emission, num_timesteps, batch_size):
    """Ancestral sample of (latents, observations), each a length-num_timesteps list of
    [batch_size, ...] tensors (or dicts), from the generative model (statistics.py:108-162)."""
    latents, observations = [], []
    for time in range(num_timesteps):
        if time == 0:
            latent_dist = initial()
        else:
            latent_dist = transition(previous_latents=latents, time=time,
                                     previous_observations=observations[:time])
        latents.append(state.sample(latent_dist, batch_size, 1))
        observations.append(state.sample(
            emission(latents=latents, time=time, previous_observations=observations[:time]),
            batch_size, 1))

    def drop_particle_dim(value):
        if isinstance(value, dict):
            return {key: drop_particle_dim(item) for key, item in value.items()}
        return value.squeeze(1)

    return [drop_particle_dim(x) for x in latents], [drop_particle_dim(y) for y in observations]
