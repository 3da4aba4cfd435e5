"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement (PyTorch-CPU + NumPy + SciPy) of the
reference's SMC / importance-sampling path, operation for operation, including the costs the
product removes: the O(T^2) history re-gather, the host round trip and per-row np.digitize loop of
the resampler, the final stack of all weight tensors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (aesmc_amd/) never does.

Pinning: oracle/capture_golden.py (run in the build container, where /root/reference is importable)
records the imported reference's inputs, random draws and outputs as tests/golden/*.npz;
tests/test_oracle.py proves this port equal to those fixtures under the replayed draws wherever
the tests run (indices exact, floats to the last place on the capture host).  The arithmetic lives in unpinned third-party libraries
(reference setup.py:59 lists torch only; numpy / scipy are implicit): fixtures were captured with
torch 2.10.0, numpy 2.2.6, scipy 1.15.3.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import enum
import warnings

import numpy as np
import scipy.special
import torch


# ---- aesmc/state.py ------------------------------------------------------------------------------
class BatchShapeMode(enum.Enum):  # state.py:6-9
    NOT_EXPANDED = 0
    BATCH_EXPANDED = 1
    FULLY_EXPANDED = 2


def set_batch_shape_mode(distribution, batch_shape_mode):  # state.py:12-17
    distribution.batch_shape_mode = batch_shape_mode
    return distribution


def get_batch_shape_mode(distribution, batch_size=None, num_particles=None):  # state.py:20-58
    if hasattr(distribution, "batch_shape_mode"):
        return distribution.batch_shape_mode
    shape = tuple(distribution.batch_shape)
    if len(shape) == 0 or shape[0] != batch_size:
        return BatchShapeMode.NOT_EXPANDED
    mode = BatchShapeMode.FULLY_EXPANDED if (len(shape) > 1 and shape[1] == num_particles) \
        else BatchShapeMode.BATCH_EXPANDED
    warnings.warn("batch_shape_mode {} inferred from batch_shape {}".format(mode, shape),
                  RuntimeWarning)
    return mode


def sample(distribution, batch_size, num_particles):  # state.py:61-111
    if isinstance(distribution, dict):
        return {k: sample(v, batch_size, num_particles) for k, v in distribution.items()}
    if isinstance(distribution, torch.Tensor):
        return distribution
    mode = get_batch_shape_mode(distribution, batch_size, num_particles)
    if not distribution.has_rsample:
        raise ValueError("distribution not reparameterizable")
    if mode == BatchShapeMode.NOT_EXPANDED:
        return distribution.rsample(sample_shape=(batch_size, num_particles))
    if mode == BatchShapeMode.BATCH_EXPANDED:
        return distribution.rsample(sample_shape=(num_particles,)).transpose(0, 1)
    return distribution.rsample(sample_shape=())


def log_prob(distribution, value):  # state.py:114-155 (dict branch of the reference is dead code)
    gap = value.ndimension() - len(distribution.event_shape) - len(distribution.batch_shape)
    if gap in (0, 2):
        distribution._validate_sample(value)
        logp = distribution.log_prob(value)
    elif gap == 1:
        logp = distribution.log_prob(value.transpose(0, 1)).transpose(0, 1)
    else:
        raise RuntimeError("incompatible batch_shape / value.shape")
    return torch.sum(logp.reshape(value.size(0), value.size(1), -1), dim=2)


def resample(value, ancestral_index):  # state.py:158-183: element-granular torch.gather
    if isinstance(value, dict):
        return {k: resample(v, ancestral_index) for k, v in value.items()}
    assert ancestral_index.size() == value.size()[:2]
    index = ancestral_index.reshape(ancestral_index.shape + (1,) * (value.dim() - 2))
    return torch.gather(value, dim=1, index=index.expand_as(value))


def expand_observation(observation, num_particles):  # state.py:186-203
    if isinstance(observation, dict):
        return {k: expand_observation(v, num_particles) for k, v in observation.items()}
    return observation.unsqueeze(1).expand(observation.size(0), num_particles,
                                           *observation.size()[1:])


# ---- aesmc/math.py -------------------------------------------------------------------------------
def lognormexp(values, dim=0):  # math.py:6-30
    if isinstance(values, np.ndarray):
        return values - scipy.special.logsumexp(values, axis=dim, keepdims=True)
    return values - torch.logsumexp(values, dim=dim, keepdim=True)


def exponentiate_and_normalize(values, dim=0):  # math.py:33-51
    out = lognormexp(values, dim=dim)
    return np.exp(out) if isinstance(out, np.ndarray) else torch.exp(out)


# ---- aesmc/inference.py --------------------------------------------------------------------------
def sample_ancestral_index(log_weight):  # inference.py:234-269
    if torch.sum(log_weight != log_weight).item() != 0:  # :244-245, one host sync per call
        raise FloatingPointError("log_weight contains nan element(s)")
    batch_size, num_particles = log_weight.size()
    uniforms = np.random.uniform(size=[batch_size, 1])                      # :250 float64
    positions = (uniforms + np.arange(0, num_particles)) / num_particles    # :251 float64
    weights = exponentiate_and_normalize(log_weight.detach().cpu().numpy(), dim=1)  # :253-254
    cdf = np.cumsum(weights, axis=1)                                        # :257, input dtype
    cdf = cdf / np.max(cdf, axis=1, keepdims=True)                          # :260-261
    out = np.zeros([batch_size, num_particles])                             # :248 float64 buffer
    for row in range(batch_size):                                           # :263-264
        out[row] = np.digitize(positions[row], cdf[row])
    return torch.from_numpy(out).long()                                     # :266-269


def get_resampled_latents(latents, ancestral_indices):  # inference.py:196-231
    assert len(ancestral_indices) == len(latents) - 1
    first = next(iter(latents[0].values())) if isinstance(latents[0], dict) else latents[0]
    batch_size, num_particles = first.size()[:2]
    lineage = torch.arange(0, num_particles).long().unsqueeze(0).expand(batch_size, num_particles)
    out = []
    for t in range(len(latents) - 1, -1, -1):
        out.insert(0, resample(latents[t], lineage))
        if t != 0:
            lineage = torch.gather(ancestral_indices[t - 1], dim=1, index=lineage)
    return out


def infer(inference_algorithm, observations, initial, transition, emission, proposal,
          num_particles, return_log_marginal_likelihood=False, return_latents=True,
          return_original_latents=False, return_log_weight=True, return_log_weights=False,
          return_ancestral_indices=False):  # inference.py:8-193
    if inference_algorithm not in ("is", "smc"):
        raise ValueError("inference_algorithm must be either is or smc")
    smc = inference_algorithm == "smc"
    first_obs = observations[0]
    batch_size = (next(iter(first_obs.values())) if isinstance(first_obs, dict) else first_obs).size(0)
    originals, indices, log_weights, history = [], [], [], []
    for t in range(len(observations)):
        if t == 0:                                                           # :85-98
            q = proposal(time=0, observations=observations)
        else:                                                                # :99-126
            if smc:
                indices.append(sample_ancestral_index(log_weights[-1]))     # :101
                parents = [resample(x, indices[-1]) for x in history]        # :102-104, O(t) gathers
            else:
                parents = history
            q = proposal(previous_latents=parents, time=t, observations=observations)
        x = sample(q, batch_size, num_particles)
        history += [x]
        lq = log_prob(q, x)
        if t == 0:
            lp = log_prob(initial(), x)
            lg = log_prob(emission(latents=history, time=0),
                          expand_observation(observations[0], num_particles))
            log_weights.append(lp + lg - lq)                                 # :97-98
        else:
            lp = log_prob(transition(previous_latents=parents, time=t,
                                     previous_observations=observations[:t]), x)
            lg = log_prob(emission(latents=history, time=t, previous_observations=observations[:t]),
                          expand_observation(observations[t], num_particles))
            log_weights.append(lp + lg - lq)                                 # :125-126
        originals.append(x)
    lml = latents = log_weight = None
    if smc:                                                                  # :128-154
        if return_log_marginal_likelihood:
            lml = torch.sum(torch.logsumexp(torch.stack(log_weights, dim=0), dim=2)
                            - np.log(num_particles), dim=0)
        if return_latents:
            latents = get_resampled_latents(originals, indices)
        if return_log_weight:
            log_weight = log_weights[-1]
    else:                                                                    # :155-186
        if return_log_marginal_likelihood or return_log_weight:
            log_weight = torch.sum(torch.stack(log_weights, dim=0), dim=0)
        if return_log_marginal_likelihood:
            lml = torch.logsumexp(log_weight, dim=1) - np.log(num_particles)
        if return_latents:
            latents = originals
        if return_original_latents or return_ancestral_indices:
            raise RuntimeWarning("flag only applicable to smc")
        if not return_log_weight:
            log_weight = None
    return {"log_marginal_likelihood": lml, "latents": latents,
            "original_latents": originals if (smc and return_original_latents) else None,
            "log_weight": log_weight,
            "log_weights": log_weights if return_log_weights else None,
            "ancestral_indices": indices if (smc and return_ancestral_indices) else None,
            "last_latent": x}


# ---- aesmc/losses.py -----------------------------------------------------------------------------
def get_loss(observations, num_particles, algorithm, initial, transition, emission, proposal):
    """losses.py:5-65."""
    result = infer({"iwae": "is", "aesmc": "smc"}[algorithm], observations, initial, transition,
                   emission, proposal, num_particles, return_log_marginal_likelihood=True,
                   return_latents=False, return_log_weight=False)
    return -torch.mean(result["log_marginal_likelihood"])
