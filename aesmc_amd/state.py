"""Particle-state helpers with the semantics of the reference's aesmc/state.py.

Every particle quantity is laid out [batch_size, num_particles, ...].  `resample` is the HIP row
gather (kernel K3); the rest is shape plumbing around `torch.distributions` objects that the
user's initial / transition / emission / proposal callables return.
"""
import enum
import warnings

import contextlib
import contextvars

import torch

from . import _kernels
from . import _ops
from . import _philox
from . import _syncfree
from . import settings
from ._lazy import LazyAffine, LazyDraw, LazyInitialDraw, LazyParticles, LazyResampled
from ._lazy import real as _lazy_real
from .linear_gaussian import AffineNormal, affine_terms

def set_validation_mode(mode):
    """How `log_prob` performs the reference's explicit `distribution._validate_sample(value)`
    (aesmc/state.py:142) for values on the HIP device:
      'deferred' (default) — shapes are checked on the host at once; the support check runs on
          the device without synchronising and a violation raises ValueError at the end of
          `inference.infer` (values outside a real-valued support are NaN and surface through the
          resampler's NaN flag as FloatingPointError);
      'eager' — call `_validate_sample` as the reference does (one host sync per call).
    (The process-wide default of `settings.Settings.validation_mode`; `settings.override` scopes it.)"""
    settings.set_default(validation_mode=mode)


def _is_real_support(support):
    constraints = torch.distributions.constraints
    while isinstance(support, constraints.independent):
        support = support.base_constraint
    return support is constraints.real or isinstance(support, type(constraints.real))


def _validate_sample(distribution, value):
    if settings.current().validation_mode == "eager" or not value.is_cuda:
        distribution._validate_sample(value)
        return
    _syncfree.validate_sample(distribution, value)      # shapes on the host at once, the support on the device


class BatchShapeMode(enum.Enum):
    """How much of [batch_size, num_particles] a distribution's batch_shape already spans
    (reference: aesmc/state.py:6-9)."""
    NOT_EXPANDED = 0    # batch_shape == [...]
    BATCH_EXPANDED = 1  # batch_shape == [batch_size, ...]
    FULLY_EXPANDED = 2  # batch_shape == [batch_size, num_particles, ...]


def set_batch_shape_mode(distribution, batch_shape_mode):
    """Tags `distribution` with an explicit BatchShapeMode and returns it (state.py:12-17)."""
    distribution.batch_shape_mode = batch_shape_mode
    return distribution


def _guess_batch_shape_mode(batch_shape, batch_size, num_particles):
    """Returns (mode, ambiguous) from the leading batch dims alone (state.py:24-58)."""
    rank = len(batch_shape)
    if rank == 0 or batch_shape[0] != batch_size:
        return BatchShapeMode.NOT_EXPANDED, False
    if rank >= 2 and batch_shape[1] == num_particles:
        return BatchShapeMode.FULLY_EXPANDED, True
    return BatchShapeMode.BATCH_EXPANDED, True


def get_batch_shape_mode(distribution, batch_size=None, num_particles=None):
    """The explicit tag if one was set, else a guess from batch_shape; a guess that rests on a
    coincidence of sizes emits a RuntimeWarning, as the reference does."""
    if hasattr(distribution, "batch_shape_mode"):
        return distribution.batch_shape_mode
    mode, ambiguous = _guess_batch_shape_mode(distribution.batch_shape, batch_size, num_particles)
    if ambiguous:
        warnings.warn(
            "Inferred batch_shape_mode ({}) of distribution ({}) might be wrong given its "
            "batch_shape ({}), batch_size ({}) and num_particles ({}). Consider specifying the "
            "batch_shape_mode explicitly.".format(mode, distribution, distribution.batch_shape,
                                                  batch_size, num_particles), RuntimeWarning)
    return mode


_SAMPLE_SHAPE = {
    BatchShapeMode.NOT_EXPANDED: lambda b, k: (b, k),
    BatchShapeMode.BATCH_EXPANDED: lambda b, k: (k,),
    BatchShapeMode.FULLY_EXPANDED: lambda b, k: (),
}


def sample(distribution, batch_size, num_particles):
    """Reparameterised draw shaped [batch_size, num_particles, ...] (state.py:61-111).

    Accepts a Distribution, a dict of them (sampled per key) or a Tensor (returned as is).
    """
    if isinstance(distribution, dict):
        return {key: sample(dist, batch_size, num_particles) for key, dist in distribution.items()}
    if isinstance(distribution, torch.Tensor):
        return distribution
    if not isinstance(distribution, torch.distributions.Distribution):
        raise AttributeError("distribution must be a dict or a torch.distributions.Distribution. "
                             "Got: {}".format(distribution))
    mode = get_batch_shape_mode(distribution, batch_size, num_particles)
    if mode not in _SAMPLE_SHAPE:
        raise ValueError("batch_shape_mode {} not supported".format(mode))
    if not distribution.has_rsample:
        raise ValueError("distribution not reparameterizable")
    sample_shape = _SAMPLE_SHAPE[mode](batch_size, num_particles)
    batch_expanded = mode == BatchShapeMode.BATCH_EXPANDED
    draw = _fused_normal_rsample(distribution, sample_shape, batch_expanded) if settings.current().fused_normal else None
    if draw is None:
        draw = distribution.rsample(sample_shape=sample_shape)
        if batch_expanded:
            draw = draw.transpose(0, 1)  # [K, B, ...] -> [B, K, ...] view, as in the reference
    return draw


def log_prob(distribution, value):
    """Log-density of `value` [batch_size, num_particles, ...] summed over everything past the
    first two dims -> [batch_size, num_particles] (state.py:114-155).

    The distribution may be missing zero, one (num_particles) or two leading batch dims.  A dict
    of distributions sums its members' log-densities (the reference's dict branch is unreachable:
    it names an undefined variable, state.py:130; this is the evident intent).
    """
    if isinstance(value, LazyParticles):
        value = value.materialise()
    if isinstance(distribution, dict):
        total = None
        for key, dist in distribution.items():
            term = log_prob(dist, value[key])
            total = term if total is None else total + term
        return total
    if not isinstance(distribution, torch.distributions.Distribution):
        raise AttributeError("distribution must be a dict or a torch.distributions.Distribution. "
                             "Got: {}".format(distribution))
    missing = (value.dim() - len(distribution.event_shape)) - len(distribution.batch_shape)
    if missing not in (0, 1, 2):
        raise RuntimeError("Incompatible distribution.batch_shape ({}) and value.shape ({}).".format(
            distribution.batch_shape, value.shape))
    if missing != 1:
        _validate_sample(distribution, value)
    fused = _fused_normal_views(distribution, value, missing) if settings.current().fused_normal else None
    if fused is not None:
        if missing == 1 and distribution._validate_args:  # what Normal.log_prob itself would check
            _validate_sample(distribution, value.transpose(0, 1))
        return _ops.normal_log_prob_sum(value, *fused)
    if missing == 1:
        logp = distribution.log_prob(value.transpose(0, 1)).transpose(0, 1)
    else:
        logp = distribution.log_prob(value)
    return logp.reshape(value.size(0), value.size(1), -1).sum(dim=2)


def normal_log_weight(prior_dist, proposal_dist, latent, emission_dist, observation, defer_grad=False):
    """One step's log-weight  log_prob(prior, latent) + log_prob(emission, observation)
    - log_prob(proposal, latent)  (aesmc/inference.py:112-126) in ONE kernel (K5) when all three
    distributions are plain Normals (any scales) on the HIP device; None otherwise — the
    caller then takes three `log_prob` calls, which give bit-identical numbers.  `observation` is
    already expanded over particles.  Validation is what the three `log_prob` calls would do.
    `defer_grad`: return (log-weights without an autograd node, K5's operands) for a caller that
    differentiates only through the row log-sum-exp (`_ops.attach_lse`)."""
    if not settings.current().fused_normal:
        return None
    if type(latent) is LazyInitialDraw and latent.is_pending:
        log_weight = _initial_step(prior_dist, proposal_dist, latent, emission_dist, observation, defer_grad)
        if log_weight is not None:
            return log_weight
        latent = latent.materialise()      # K6, as `sample` would have drawn it; the three-launch route below
    if type(latent) is LazyDraw and latent.is_pending and getattr(latent, "wide", False):
        log_weight = _wide_step(prior_dist, proposal_dist, latent, emission_dist, observation, defer_grad)
        if log_weight is not None:
            return log_weight
    affine = _affine_step_operands(prior_dist, proposal_dist, latent, emission_dist, observation)
    if type(latent) is LazyDraw and latent.is_pending:
        # a deferred draw: K16 / K15 forms it together with the log-weight when the step is linear-Gaussian in this
        # very proposal and the log-weights need no autograd node of their own; else K9 forms it now
        fused = affine is not None and affine.is_draw and (
            defer_grad or not (torch.is_grad_enabled() and affine.requires_grad()))
        if fused:
            # (a previous latent that nothing has gathered yet is fetched through the ancestors by this launch)
            x_t = torch.empty(latent.shape, dtype=latent.dtype, device=latent.device)
            log_weight = _ops.affine_propagate(affine.with_latent(x_t), latent.noise)
            latent.resolve(x_t)
            affine = affine.with_latent(x_t)
            # (what `_validate_sample` would check is settled: the three are Normals — real support — whose batch
            #  shapes `_affine_step_operands` matched against the very tensors the launch wrote and read)
            return (log_weight, affine) if defer_grad else log_weight
        x_t = latent.materialise()
        if affine is not None:
            affine = affine.with_latent(x_t)
        latent = x_t
    elif isinstance(latent, LazyParticles):
        latent = latent.materialise()
        if affine is not None:
            affine = affine.with_latent(latent)
    if affine is not None:
        affine = affine.gathered()      # every other route reads x_{t-1}[ancestors] as a tensor
        for distribution, value in ((prior_dist, latent), (emission_dist, observation), (proposal_dist, latent)):
            _validate_sample(distribution, value)
        if defer_grad:
            return _ops.affine_log_weight_deferred(affine)
        return _ops.affine_log_weight(affine)
    operands = []
    for distribution, value in ((prior_dist, latent), (emission_dist, observation),
                                (proposal_dist, latent)):
        if not isinstance(distribution, torch.distributions.Distribution) or not torch.is_tensor(value):
            return None
        missing = (value.dim() - len(distribution.event_shape)) - len(distribution.batch_shape)
        if missing not in (0, 1, 2):
            return None  # let log_prob raise its RuntimeError
        views = _fused_normal_views(distribution, value, missing)
        if views is None:
            return None
        operands.append((distribution, value, missing, views))
    for distribution, value, missing, _ in operands:
        if missing != 1:
            _validate_sample(distribution, value)
        elif distribution._validate_args:
            _validate_sample(distribution, value.transpose(0, 1))
    (_, _, _, (loc_p, scale_p)), (_, _, _, (loc_g, scale_g)), (_, _, _, (loc_q, scale_q)) = operands
    if defer_grad:
        return _ops.normal_log_weight_deferred(latent, loc_p, scale_p, observation, loc_g, scale_g, loc_q, scale_q)
    return _ops.normal_log_weight(latent, loc_p, scale_p, observation, loc_g, scale_g, loc_q, scale_q)


# set by `infer` for its own duration: only there is a deferred draw guaranteed to be filled before use
_DEFER_DRAWS = contextvars.ContextVar("aesmc_amd_defer_draws", default=False)


@contextlib.contextmanager
def deferring_draws():
    token = _DEFER_DRAWS.set(True)
    try:
        yield
    finally:
        _DEFER_DRAWS.reset(token)


_TORCH_STANDARD_NORMAL = torch.distributions.normal._standard_normal
def set_kernel_noise(enabled):
    """On (default): a deferred draw's float32 noise is formed inside the propagation launch from PyTorch's own
    Philox stream (same values, same generator state afterwards as `_standard_normal`); off: `_standard_normal`
    materialises it first (rounds 1-2).  (The process-wide default of `settings.Settings.kernel_noise`.)"""
    settings.set_default(kernel_noise=bool(enabled))


def _kernel_noise_applies(source):
    """Float32 on the HIP device, PyTorch's `_standard_normal` in place (tests replay recorded noise through
    it), and — inside a hipGraph capture — a `_philox.GraphNoise` scope to hold the generator state on the device
    (graphs.GraphedLoss opens one when every draw of the ELBO is one this package can place itself)."""
    return (settings.current().kernel_noise and source.is_cuda and source.dtype == torch.float32 and _kernels.get().name == "hip" and
            torch.distributions.normal._standard_normal is _TORCH_STANDARD_NORMAL and
            (_philox.graph_noise() is not None or not torch.cuda.is_current_stream_capturing()) and
            _philox.verified(source.device))


def _standard_normal(shape, dtype, device):
    """`torch.distributions.normal._standard_normal` as `Normal.rsample` calls it (aesmc/state.py:98) — except
    inside a hipGraph capture that holds the generator state itself (`_philox.GraphNoise`): there the same values
    come from aesmc_philox_normal_fill, so that every draw of the captured region sits at the offset the eager
    evaluation gives it.  Outside a capture the call is only counted (see `_philox.COUNTERS`)."""
    draw = torch.distributions.normal._standard_normal
    if draw is _TORCH_STANDARD_NORMAL and dtype == torch.float32 and device.type == "cuda" and \
            settings.current().kernel_noise and \
            _kernels.get().name == "hip" and _philox.verified(device):
        numel = 1
        for size in shape:
            numel *= size
        if numel > 0:
            if _philox.graph_noise() is not None and torch.cuda.is_current_stream_capturing():
                return _kernels.get().philox_normal(_philox.reserve(numel, device), tuple(shape), device)
            _philox.COUNTERS["replaceable"] += _philox.consumed(numel, _philox.launch_threads(numel, device))
    return draw(shape, dtype=dtype, device=device)


def _noise_tensor(noise, latent):
    """The noise of a deferred draw as a tensor: what was drawn, or — for a reservation — what PyTorch would
    have drawn (aesmc_philox_normal_fill)."""
    if isinstance(noise, _philox.NoiseStream):
        return _kernels.get().philox_normal(noise, tuple(latent.shape), latent.device)
    return noise


def materialise_draw(latent):
    """The latent with its values: a draw that was left to the launch that weighs the step (`LazyDraw`) and that no
    such launch has formed is drawn now (K9, differentiable); anything else passes through.  Dict-aware."""
    if isinstance(latent, dict):
        return {key: materialise_draw(value) for key, value in latent.items()}
    return latent.materialise() if isinstance(latent, LazyParticles) else latent


def _same_tensor(a, b):
    if a is b:
        return True
    if isinstance(a, LazyParticles) or isinstance(b, LazyParticles):
        return False
    return _same_storage_and_history(a, b)


def _same_storage_and_history(a, b):
    """One operand for the fused step: the same storage view AND the same autograd identity — `x` and
    `x.detach()` (a stop-gradient into one callable) share memory but not gradients, and the fused backward
    has one gradient slot per operand."""
    if a is b:
        return True
    return (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
            and a.requires_grad == b.requires_grad and a.grad_fn is b.grad_fn)


def _affine_step_operands(prior_dist, proposal_dist, latent, emission_dist, observation):
    """K10's operands when the step is linear-Gaussian — transition and proposal linear-Gaussian terms
    (`linear_gaussian.affine_terms`: an AffineNormal, or a plain Normal whose location is a recorded
    `x @ W.t() + c`) in the SAME previous latent, the emission one in the latent being weighed, one scale value
    each, the observation one row per batch element expanded over particles — else None."""
    prior, proposal, emission = affine_terms(prior_dist), affine_terms(proposal_dist), affine_terms(emission_dist)
    if prior is None or proposal is None or emission is None:
        return None
    if not (torch.is_tensor(latent) and torch.is_tensor(observation) and latent.dim() == 3 and
            observation.dim() == 3 and observation.stride(1) == 0):
        return None
    x_prev = prior.source
    if not (_same_tensor(proposal.source, x_prev) and _same_tensor(emission.source, latent)):
        return None
    if type(x_prev) is LazyDraw or type(x_prev) is LazyAffine:
        return None
    pending = None
    if type(x_prev) is LazyResampled:
        pending = x_prev.pending        # (x_{t-1}, ancestors) while nothing has gathered them
        if pending is None:
            x_prev = x_prev.materialise()
    y_rows = observation[:, 0]
    transition_map = (prior.weight, prior.offset)
    emission_map = (emission.weight, emission.offset)
    proposal_map = (proposal.weight, proposal.offset)
    scales = (prior.scale_param, emission.scale_param, proposal.scale_param)
    if not _kernels.get().affine_logweight_covers(x_prev, latent, y_rows, transition_map, emission_map, proposal_map,
                                                  scales):
        return None
    operands = _ops.AffineOperands((x_prev, latent, y_rows, transition_map[0], transition_map[1], emission_map[0],
                                    emission_map[1], proposal_map[0], proposal_map[1]) + scales)
    # the latent is this very proposal's reparameterised draw: the step can be one autograd node
    draw_of = latent.terms if type(latent) is LazyDraw else getattr(latent, "_aesmc_draw_of", None)
    operands.is_draw = draw_of is not None and draw_of.same_terms(proposal)
    operands.pending_gather = pending
    return operands


def _initial_step(prior_dist, proposal_dist, latent, emission_dist, observation, defer_grad=False):
    """The FIRST timestep in one launch (K20, aesmc_affine_normal_initial_step) when `latent` is this proposal's pending
    transposed draw (`LazyInitialDraw`: a BATCH_EXPANDED Normal), the prior a plain Normal whose parameters do not vary
    along the particles, and the emission linear-Gaussian in the latent itself (an AffineNormal, or `Normal(x @ C.t() +
    g, s)` recorded on the lazy draw) with one scale value: the draw, the emission's location and the three log-densities
    — the bits of K6 + K8 + K5, the route taken otherwise (None).  Under autograd only for a caller that differentiates
    the log-weights through their row log-sum-exp (`defer_grad`): the draw resolves to a tensor with `_NormalRsample`'s
    history, and the operands come back for `_ops.attach_lse` (`_ops.InitialOperands`), whose backward forms the
    emission's location again (K8) in front of K5's backward."""
    provider = _kernels.get()
    if provider.name != "hip" or not hasattr(provider, "affine_initial_step") or latent.distribution is not proposal_dist:
        return None
    if len(latent.shape) != 3 or latent.dtype != torch.float32:
        return None
    if not (torch.is_tensor(observation) and observation.dim() == 3 and observation.stride(1) == 0 and
            observation.dtype == torch.float32 and observation.device == latent.device):
        return None
    emission = affine_terms(emission_dist)
    if emission is None or emission.source is not latent:
        return None
    weight, offset, scale_g = emission.weight, emission.offset, emission.scale_param
    B, K, dx = latent.shape
    if not (torch.is_tensor(weight) and weight.dim() == 2 and weight.size(1) == dx and 1 <= dx <= 16 and
            1 <= weight.size(0) <= 16 and weight.dtype == torch.float32 and weight.device == latent.device and
            tuple(observation.shape) == (B, K, weight.size(0))):
        return None
    if not (torch.is_tensor(scale_g) and scale_g.numel() == 1 and scale_g.dtype == torch.float32 and
            scale_g.device == latent.device):
        return None
    if offset is not None and not (torch.is_tensor(offset) and offset.dtype == torch.float32 and
                                   offset.device == latent.device and tuple(offset.shape) in ((weight.size(0),),
                                                                                              (B, weight.size(0)))):
        return None
    views = []
    for distribution in (prior_dist, proposal_dist):
        if not isinstance(distribution, torch.distributions.Distribution):
            return None
        missing = (3 - len(distribution.event_shape)) - len(distribution.batch_shape)
        if missing not in (0, 1, 2) or (distribution is proposal_dist and missing != 1):
            return None
        pair = _fused_normal_views(distribution, latent, missing)
        if pair is None or any(K > 1 and view.stride(1) != 0 for view in pair):
            return None
        views.append(pair)
    (loc_p, scale_p), (loc_q, scale_q) = views
    scale_g_view = scale_g.expand(B, K, weight.size(0))      # (as `AffineNormal.scale` expands it: the same reduction folds its gradient)
    operands = (loc_p, scale_p, observation, weight, offset, scale_g_view, loc_q, scale_q)
    wants_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in operands)
    if wants_grad and not defer_grad:
        return None      # differentiable log-weights themselves: the three launches' own autograd nodes
    x_0 = torch.empty((B, K, dx), dtype=torch.float32, device=latent.device)
    log_weight = provider.affine_initial_step(latent.noise, loc_q.detach(), scale_q.detach(), loc_p.detach(), scale_p.detach(),
                                              observation.detach(), weight.detach(),
                                              None if offset is None else offset.detach(), scale_g_view.detach(), x_0)
    if log_weight is None:
        return None
    # (what `_validate_sample` would check is settled as in the fused step: the three are Normals — real support —
    #  whose parameter shapes were matched against the very tensors the launch wrote and read)
    if not wants_grad:
        latent.resolve(x_0)
        return (log_weight, ()) if defer_grad else log_weight
    # (the draw's own views of the proposal's parameters — expanded to the noise's shape, then transposed — as K6's node
    #  would have been handed: the reductions that fold their gradients are then the same launches, the same bits)
    x_0 = _ops.normal_rsample_given(latent.noise.transpose(0, 1), latent.loc.transpose(0, 1), latent.scale.transpose(0, 1), x_0)
    latent.resolve(x_0)
    port = _ops.particle_affine_port(x_0, weight, offset)
    return log_weight, _ops.InitialOperands((x_0, loc_p, scale_p, observation, port, weight.detach(),
                                             None if offset is None else offset.detach(), scale_g_view, loc_q, scale_q))


def _wide_step(prior_dist, proposal_dist, latent, emission_dist, observation, defer_grad=False):
    """A linear-Gaussian step on rows of 128 values whose latent is this proposal's deferred draw: K17 + K18 form the draw
    (through the ancestors when nothing has gathered x_{t-1} yet) and the log-weights; the draw resolves to the tensor
    they wrote.  Under autograd only for a caller that differentiates the log-weights through their row log-sum-exp
    (`defer_grad`: what `get_loss` asks of `infer`): the launches run without autograd nodes and the step's operands come
    back beside the log-weights — `_ops.affine_step` ties x_t and the log-sum-exp to them, and the backward RECOMPUTES the
    locations from x_{t-1}, the ancestors and x_t (`_kernels.affine_step_backward_wide`): nothing but x_t, the
    log-weights and the indices is kept per timestep.  None when the step is not of that kind (the other routes apply)."""
    if not (torch.is_tensor(observation) and observation.dim() == 3 and observation.stride(1) == 0):
        return None
    if torch.is_grad_enabled() and not defer_grad:
        prior, proposal, emission = affine_terms(prior_dist), affine_terms(proposal_dist), affine_terms(emission_dist)
        if any(t is not None and any(v is not None and torch.is_tensor(v) and v.requires_grad
                                     for v in (t.source, t.weight, t.offset, t.scale_param))
               for t in (prior, proposal, emission)):
            return None      # differentiable log-weights themselves: the GEMM route's autograd
    prior, proposal, emission = affine_terms(prior_dist), affine_terms(proposal_dist), affine_terms(emission_dist)
    if prior is None or proposal is None or emission is None or not latent.terms.same_terms(proposal):
        return None
    x_prev = prior.source
    if not (_same_tensor(proposal.source, x_prev) and _same_tensor(emission.source, latent)):
        return None
    if type(x_prev) in (LazyDraw, LazyAffine):
        return None
    provider = _kernels.get()
    scales = (prior.scale_param, emission.scale_param, proposal.scale_param)
    if not (provider.affine_wide_covers(x_prev, prior.weight, prior.offset, scales[0]) and
            provider.affine_wide_covers(latent, emission.weight, emission.offset, scales[1]) and
            provider.affine_wide_covers(x_prev, proposal.weight, proposal.offset, scales[2])):
        return None
    ancestors = None
    if type(x_prev) is LazyResampled:
        if x_prev.pending is not None:
            x_prev, ancestors = x_prev.pending        # (x_{t-1}, ancestors): fetched inside the launch
        else:
            x_prev = x_prev.materialise()
    y_rows = observation[:, 0]
    if y_rows.dtype != torch.float32 or y_rows.device != latent.device or y_rows.dim() != 2 or \
            tuple(y_rows.shape) != (latent.shape[0], emission.weight.size(0)):
        return None      # (e.g. a float64 observation against a float32 model: PyTorch's promotion, not K18's bytes)
    x_t = torch.empty(latent.shape, dtype=torch.float32, device=latent.device)
    maps = ((prior.weight.detach(), prior.offset), (emission.weight.detach(), emission.offset),
            (proposal.weight.detach(), proposal.offset))
    log_weight = provider.affine_propagate_wide(x_prev.detach(), latent.noise, y_rows, *maps, scales, x_t, ancestors=ancestors)
    if log_weight is None and not torch.is_tensor(latent.noise):
        # the launch cannot form this shape's noise itself: the same values as a tensor (the fill kernel), once
        latent.noise = _noise_tensor(latent.noise, latent)
        log_weight = provider.affine_propagate_wide(x_prev.detach(), latent.noise, y_rows, *maps, scales, x_t,
                                                    ancestors=ancestors)
    if log_weight is None:
        return None
    latent.resolve(x_t)
    if not defer_grad:
        return log_weight
    # slot 0: x_{t-1} as the step read it — the pending LazyResampled itself while its rows were fetched through the
    # ancestors (`_ops.affine_step` takes (source, ancestors) out of `pending_gather`), else the tensor
    first = prior.source if ancestors is not None else x_prev
    operands = _ops.AffineOperands((first, x_t, y_rows, prior.weight, prior.offset, emission.weight, emission.offset,
                                    proposal.weight, proposal.offset) + scales)
    operands.is_draw, operands.wide = True, True
    operands.pending_gather = (x_prev, ancestors) if ancestors is not None else None
    return log_weight, operands


def set_fused_normal(enabled):
    """Switches the fused Normal kernels (log-density K4 / K5 inside `log_prob`, draw K6 inside
    `sample`) on (default) or off; off evaluates `distribution.log_prob` / `rsample` in eager
    PyTorch exactly as the reference does.  (The process-wide default of `settings.Settings.fused_normal`.)"""
    settings.set_default(fused_normal=bool(enabled))


def _fused_normal_rsample(distribution, sample_shape, swap_leading_dims):
    """`distribution.rsample(sample_shape)` for a plain Normal (optionally inside Independent) with
    tensor parameters on the HIP device: the noise comes from the very call torch makes
    (torch/distributions/normal.py rsample -> `_standard_normal`, so the generator advances
    identically) and `loc + eps * scale` is one pass of kernel K6 instead of two eager ones.
    With `swap_leading_dims` (a BATCH_EXPANDED distribution drawn as [K, B, ...]) the result is
    the transposed draw, [B, K, ...] — the same values the reference's `.transpose(0, 1)` view
    holds (state.py:102-103), stored densely.  None for anything else."""
    base = distribution
    if type(base) is torch.distributions.Independent:
        base = base.base_dist
    terms = affine_terms(base) if len(sample_shape) == 0 and not swap_leading_dims else None
    if terms is not None:
        # a linear-Gaussian term (AffineNormal, or Normal(x @ W.t() + c, s) recorded on a lazy latent): location +
        # noise in one pass (K9), or — inside `infer` — left to the launch that weighs the step (K16 / K15)
        scale = terms.scale_param
        if scale.numel() == 1 and scale.dtype == terms.source.dtype and scale.device == terms.source.device and \
                _kernels.get().affine_covers(terms.source, terms.weight, terms.offset):
            source = terms.source
            if _DEFER_DRAWS.get() and getattr(terms, "defer_draw", None) is not False:
                if _kernel_noise_applies(source):
                    # the draw's NOISE is left to that launch too: reserve, in PyTorch's own generator, exactly what
                    # `_standard_normal` would have consumed here — the kernel forms the same values from (seed,
                    # offset) — so the stream and everything drawn after it are unchanged
                    eps = _philox.reserve(source.numel() // source.size(-1) * terms.weight.size(0), source.device)
                else:
                    eps = _standard_normal(base.batch_shape, dtype=source.dtype, device=source.device)
                # holds no values: whoever reads them gets the draw (K9); a model that only describes distributions
                # in terms of it — `Normal(latents[-1] @ C.t(), s)` — never does
                return LazyDraw(terms, eps)
            eps = _standard_normal(base.batch_shape, dtype=source.dtype, device=source.device)
            draw = _ops.affine_rsample(source, terms.weight, terms.offset, scale, eps)
            draw._aesmc_draw_of = terms      # lets `infer` differentiate the whole step in one node (K14)
            return draw
        if _DEFER_DRAWS.get() and getattr(terms, "defer_draw", None) is not False and \
                _kernels.get().name == "hip" and _kernels.get().affine_wide_covers(terms.source, terms.weight, terms.offset,
                                                                                 scale):
            # rows of 128 values (BASELINE.json configs[4]): the draw is left to the launch that weighs the step (K17:
            # both maps of x_{t-1} on the matrix cores, the draw, two of the three densities); its noise is drawn here,
            # by the very call `rsample` makes.  (Under autograd too: `_wide_step` takes the step when the caller
            # differentiates through the row log-sum-exp only; else the draw is formed, differentiably, when it is read.)
            source = terms.source
            if _kernel_noise_applies(source):      # (reserved in PyTorch's generator; K17 forms it, or the fill kernel does)
                eps = _philox.reserve(source.numel() // source.size(-1) * terms.weight.size(0), source.device)
            else:
                eps = _standard_normal(base.batch_shape, dtype=source.dtype, device=source.device)
            draw = LazyDraw(terms, eps)
            draw.wide = True
            return draw
    if type(base) not in (torch.distributions.Normal, AffineNormal):
        return None
    loc, scale = _lazy_real(base.loc), base.scale
    if not (loc.is_cuda and scale.device == loc.device and loc.dtype == scale.dtype and
            loc.dtype in (torch.float32, torch.float64)):
        return None
    shape = base._extended_shape(torch.Size(sample_shape))
    if len(shape) < 2:
        return None
    if loc.shape != shape:
        loc = loc.expand(shape)
    if scale.shape != shape:
        scale = scale.expand(shape)
    if not swap_leading_dims and loc.dtype == torch.float32 and _kernel_noise_applies(loc) and \
            _kernels.get().RSAMPLE_DRAWN_MIN_ELEMENTS <= loc.numel() < (1 << 32) and \
            not (torch.is_grad_enabled() and scale.requires_grad):
        # the noise is formed where it is used: reserved in PyTorch's generator (the stream moves as `normal_` would
        # move it), drawn by the launch that adds loc — the same values, no noise tensor
        noise = _philox.reserve(loc.numel(), loc.device)
        draw = _ops.normal_rsample_drawn(noise, loc, scale, shape)
        if draw is not None:
            return draw
        eps = _kernels.get().philox_normal(noise, tuple(shape), loc.device)      # (declined: the values as a tensor)
    else:
        eps = _standard_normal(shape, dtype=loc.dtype, device=loc.device)
    if swap_leading_dims and _DEFER_DRAWS.get() and len(shape) == 3 and loc.dtype == torch.float32 and \
            settings.current().initial_step and _kernels.get().name == "hip":
        # inside `infer`: the first timestep's draw is left to the launch that weighs the step (K20, `_initial_step`) —
        # its noise is drawn here, by the very call `rsample` makes; whoever reads the values first gets K6's draw
        return LazyInitialDraw(distribution, loc, scale, eps)
    if swap_leading_dims:
        eps, loc, scale = eps.transpose(0, 1), loc.transpose(0, 1), scale.transpose(0, 1)
    return _ops.normal_rsample(eps, loc, scale)


def _fused_normal_views(distribution, value, missing):
    """(loc, scale) as views of value's shape if `distribution` is a plain Normal (optionally
    wrapped in Independent) that kernel K4 can evaluate on `value`; None sends the caller down
    the generic path.  `missing` leading batch dims are filled in as broadcast (stride-0) dims:
    two in front (NOT_EXPANDED) or the particle dim (BATCH_EXPANDED, where the reference
    transposes instead, state.py:144-145)."""
    base = distribution
    if type(base) is torch.distributions.Independent:
        base = base.base_dist
    if type(base) not in (torch.distributions.Normal, AffineNormal):   # AffineNormal: .loc materialises (K8)
        return None
    if not (torch.is_tensor(value) and value.is_cuda and value.dim() >= 2 and
            value.dtype in (torch.float32, torch.float64)):
        return None
    views = []
    for param in (_lazy_real(base.loc), base.scale):
        if param.dtype != value.dtype:
            return None
        if param.device != value.device:
            if param.numel() != 1:
                return None
            # Normal(0.0, 1.0): Python-number parameters live on the host.  Their value as a cached device constant
            # (read on the host — free for a host tensor — no copy per call; a capture can hold it)
            if param.requires_grad:
                param = param.to(value.device)
            else:
                param = _syncfree.constant(param.item(), param.dtype, value.device)
        if missing == 1:
            if param.dim() == 0:
                return None
            param = param.unsqueeze(1)
        if param.shape != value.shape:      # an expand of the full shape would be a no-op view
            try:
                param = param.expand(value.shape)
            except RuntimeError:
                return None
        views.append(param)
    return views


def resample(value, ancestral_index):
    """out[b, k, ...] = value[b, ancestral_index[b, k], ...] without side effects
    (state.py:158-183); recurses through dicts; differentiable w.r.t. `value`."""
    if isinstance(value, dict):
        return {key: resample(item, ancestral_index) for key, item in value.items()}
    if not torch.is_tensor(value):
        raise AttributeError("value must be a dict or a torch.Tensor. Got: {}".format(value))
    if isinstance(value, LazyParticles):
        value = value.materialise()
    assert ancestral_index.size() == value.size()[:2]
    return _ops.resample_gather(value, ancestral_index)


def expand_observation(observation, num_particles):
    """[batch_size, ...] -> stride-0 view [batch_size, num_particles, ...]; dict-aware
    (state.py:186-203)."""
    if isinstance(observation, dict):
        return {key: expand_observation(item, num_particles) for key, item in observation.items()}
    shape = list(observation.size())
    return observation.unsqueeze(1).expand(shape[0], num_particles, *shape[1:])
