"""Importance sampling / sequential Monte Carlo over a state-space model given by four callables,
with the interface of the reference's aesmc/inference.py and the per-timestep work on MI355X:

    propagate (user callables, PyTorch)  ->  K1 fused log-weight + log-sum-exp
        ->  K2 systematic ancestor indices (CDF prefix scan)  ->  K3 resample row gather

Differences from the reference that a caller can observe (all documented in DESIGN.md):
  * `previous_latents` handed to proposal / transition is a lazy read-only sequence that gathers
    an entry on first access (the reference re-gathers the whole history every step, O(T^2));
    `set_history_mode("eager")` restores plain lists;
  * data-dependent failures (NaN log-weights, an all -inf row) are detected on the device and
    raised once at the end of `infer` instead of synchronising the device every timestep.
"""
import collections.abc
import contextlib
import contextvars

import numpy as np
import torch

from . import _kernels
from . import _lazy
from . import _ops
from . import _syncfree
from . import settings
from . import state


def set_history_mode(mode):
    """'lazy' (default): resample history entries on access; 'eager': build the full list of
    re-indexed latents every step exactly as aesmc/inference.py:102-104 does.
    (The process-wide default of `settings.Settings.history_mode`; `settings.override` scopes it.)"""
    settings.set_default(history_mode=mode)


def set_lazy_gather(enabled):
    """On (default): the newest latent is handed to the callables un-gathered (`_lazy.LazyResampled`) and a
    linear-Gaussian step fetches its rows through the ancestor indices inside the launch that weighs it.
    Off: the resampling launch always re-indexes the newest latent (rounds 1-2)."""
    settings.set_default(lazy_gather=bool(enabled))


def lazy_gather(enabled):
    """`set_lazy_gather(enabled)` for the duration of a `with` block, in this context only.  With it off the callables
    are handed plain tensors, so a model written with PyTorch arithmetic (`x @ W.t() + c`) is evaluated by PyTorch itself
    — what the tests use as the independent statement of such a model."""
    return settings.override(lazy_gather=bool(enabled))


def fold_gather_backward(enabled):
    """On (default): consecutive linear-Gaussian steps hand torch.gather's backward from autograd node to autograd
    node (`_ops.StepLink`; only when `infer` returns the latents to nobody).  Off: each step's backward sums the
    children's gradients into their ancestors in a launch of its own (rounds 2-3a) — the tests' comparison run.
    A `with` block, scoped to this context."""
    return settings.override(fold_gather_backward=bool(enabled))


class ResampledHistory(collections.abc.Sequence):
    """Read-only view of [resample(x, index) for x in latents] whose entries are gathered (K3)
    when first read.  Markov models read only [-1], so a step costs one gather, not `time`."""

    def __init__(self, latents, index, newest=None, lazy_newest=False):
        self._latents = list(latents)
        self._index = index
        # `newest`: the last entry already re-indexed (the fused step copies it along the way);
        # `lazy_newest`: hand the last entry out as a LazyResampled — gathered only if something reads its
        # values (a linear-Gaussian step fetches the rows through the ancestors inside its own launch)
        self._cache = {} if newest is None else {len(self._latents) - 1: newest}
        if newest is None and lazy_newest and self._latents:
            self._cache[len(self._latents) - 1] = _lazy.LazyResampled(self._latents[-1], index)

    def newest_was_read(self):
        """Did anything need the VALUES of the lazily resampled newest entry?"""
        entry = self._cache.get(len(self._latents) - 1)
        return type(entry) is _lazy.LazyResampled and entry.pending is None

    def __len__(self):
        return len(self._latents)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return [self[i] for i in range(*item.indices(len(self._latents)))]
        position = item + len(self._latents) if item < 0 else item
        if not 0 <= position < len(self._latents):
            raise IndexError("list index out of range")
        if position not in self._cache:
            self._cache[position] = state.resample(self._latents[position], self._index)
        return self._cache[position]


_ZERO_COPY_UNIFORMS = settings.knob("AESMC_ZERO_COPY_UNIFORMS", "1") != "0"      # measurement knob


class _MappedBlock:
    """`__cuda_array_interface__` of a pinned host block at the address the device maps it to."""

    def __init__(self, address, shape, keepalive):
        self.keepalive = keepalive
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(address), False),
                                         "version": 2, "strides": None}


def _mapped_view(pinned, device):
    """A device tensor that IS the pinned float64 block `pinned` (no copy), or None when the device cannot address it
    (aesmc_host_device_pointer) or this PyTorch cannot wrap the address."""
    import ctypes
    from . import _lib
    try:
        address = ctypes.c_void_p()
        status = _lib.load().aesmc_host_device_pointer(ctypes.c_void_p(pinned.data_ptr()), ctypes.byref(address))
        if status != 0 or not address.value:
            return None
        view = torch.as_tensor(_MappedBlock(address.value, pinned.shape, pinned), device=device)
        if view.data_ptr() != address.value or view.dtype != torch.float64 or tuple(view.shape) != tuple(pinned.shape):
            return None
        view._aesmc_pinned = pinned      # the device view must not outlive the host block
        return view
    except Exception as error:      # (an optional fast path: a PyTorch that cannot wrap the address takes the copy route)
        global _MAPPED_VIEW_WARNED
        if not _MAPPED_VIEW_WARNED:
            _MAPPED_VIEW_WARNED = True
            import warnings
            warnings.warn("aesmc_amd: the pinned uniform block could not be addressed from the device ({}: {}); every "
                          "resampling step copies its uniforms instead (slower by one small launch per timestep; said "
                          "once)".format(type(error).__name__, error), RuntimeWarning)
        return None


_MAPPED_VIEW_WARNED = False


class _UniformFeed:
    """Per-resample uniforms: drawn from numpy's global RandomState one [batch_size, 1] block per
    timestep (the reference's RNG consumption, aesmc/inference.py:250) and shipped to the device
    through a pinned ring so the copy never blocks the host."""

    def __init__(self, batch_size, num_draws, device):
        self.device = device
        self.batch_size = batch_size
        self.cursor = 0
        self.mapped_rows = None
        if device.type == "cuda":
            self.host = torch.empty((max(num_draws, 1), batch_size), dtype=torch.float64,
                                    pin_memory=True)
            self.host_np = self.host.numpy()      # the same pinned memory: a draw is written straight into it
            # the resampling launch reads its 8 bytes per row straight out of that pinned block when the device can
            # address it (one slot per timestep, written once before the launch that reads it): no copy launch per step
            mapped = _mapped_view(self.host, device) if _ZERO_COPY_UNIFORMS else None
            if mapped is not None:
                self.mapped_rows = list(mapped.unbind(0))
                self.dev = mapped
            else:
                self.dev = torch.empty((max(num_draws, 1), batch_size), dtype=torch.float64,
                                       device=device)
                self.host_rows = list(self.host.unbind(0))
                self.dev_rows = list(self.dev.unbind(0))
        else:
            self.host = self.dev = None

    def next(self):
        draw = draw_uniform_block(self.batch_size)
        if self.host is None:
            return torch.from_numpy(draw)
        slot = self.cursor % len(self.host_np)
        self.cursor += 1
        self.host_np[slot] = draw
        if self.mapped_rows is not None:
            return self.mapped_rows[slot]
        row = self.dev_rows[slot]
        row.copy_(self.host_rows[slot], non_blocking=True)
        return row


def draw_uniform_block(batch_size):
    """One resampling step's uniforms, float64 [batch_size], consuming numpy's global RandomState
    exactly as aesmc/inference.py:250 does (`np.random.uniform(size=[batch_size, 1])`); inside
    `distributed.shard_scope` the block is drawn for the global batch and cut to this rank's rows."""
    from . import distributed
    shard = distributed.active_shard()
    if shard is None:
        return np.random.uniform(size=[batch_size, 1]).reshape(-1)
    global_batch, lo, hi = shard
    assert hi - lo == batch_size, "shard_scope does not match the local batch"
    return np.random.uniform(size=[global_batch, 1])[lo:hi].reshape(-1)


# set by graphs.GraphedLoss while it captures or replays eagerly: a feed with static buffers.  Per
# thread / per context: a capture in one thread does not hijack the uniforms of an infer() in another.
_FEED_OVERRIDE = contextvars.ContextVar("aesmc_amd_feed_override", default=None)


@contextlib.contextmanager
def uniform_feed(feed):
    """While active, `infer` takes its per-resample uniforms from `feed` (an object with `next()`
    returning a float64 [batch_size] device tensor) instead of drawing them itself."""
    token = _FEED_OVERRIDE.set(feed)
    try:
        yield
    finally:
        _FEED_OVERRIDE.reset(token)


def check_device_status(device=None):
    """Synchronising read-and-clear of the device status word: raises what `infer` would raise for
    anything the kernels flagged since the last check (NaN log-weights -> FloatingPointError, a row
    without a finite maximum or an ancestor index outside [0, num_particles) -> RuntimeError, a value
    outside a distribution's support -> ValueError).  `infer` and `sample_ancestral_index` call it
    themselves; the stand-alone `get_resampled_latents` / `state.resample` / `state.log_prob` only
    SET flags (they never synchronise): call this after them, or let the next `infer` report."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    _raise_for_flags(_kernels.get().read_flags(device))


def _raise_for_flags(flags):
    from . import _lib
    if flags & _lib.FLAG_INVALID_PARAMETER:      # the cause first: an invalid scale also makes the log-weights NaN
        from . import _syncfree
        raise ValueError("Expected the parameters of a distribution to satisfy their constraints, but found invalid "
                         "values (detected on the device; checks deferred lately: {})".format(
                             "; ".join(_syncfree.checked_parameters()) or "none recorded"))
    if flags & _lib.FLAG_NAN_LOG_WEIGHT:
        raise FloatingPointError("log_weight contains nan element(s)")
    if flags & _lib.FLAG_VALUE_OUTSIDE_SUPPORT:
        raise ValueError("The value argument must be within the support of the distribution "
                         "(detected on the device during log_prob)")
    if flags & _lib.FLAG_UNSORTED_INDEX:
        raise RuntimeError("aesmc_amd internal error: an index tensor tagged as sorted was not")
    if flags & (_lib.FLAG_DEGENERATE_ROW | _lib.FLAG_INDEX_OUT_OF_RANGE):
        raise RuntimeError(
            "ancestral index out of range: a row of log-weights had no finite maximum (all -inf, "
            "or +inf present), or an index outside [0, num_particles) was passed to resample")


def _first_tensor(value):
    return next(iter(value.values())) if isinstance(value, dict) else value


def sample_ancestral_index(log_weight):
    """Systematic resampling (aesmc/inference.py:234-269): log_weight [batch_size, num_particles]
    unnormalised -> zero-indexed ancestor LongTensor of the same shape on the same device.
    Draws one np.random.uniform(size=[batch_size, 1]) block; raises FloatingPointError on NaN."""
    batch_size = log_weight.size(0)
    uniforms = torch.from_numpy(np.random.uniform(size=[batch_size, 1]).reshape(-1))
    if log_weight.is_cuda:
        uniforms = uniforms.to(log_weight.device)
    index = _ops.ancestor_index(log_weight, uniforms)
    # every bit raises — also one left behind by an earlier deferred check: an all -inf (or +inf) row
    # yields indices equal to num_particles, where the reference fails inside np.digitize / torch.gather
    _raise_for_flags(_kernels.get().read_flags(log_weight.device))
    return index


def get_resampled_latents(latents, ancestral_indices):
    """Traces every final particle's genealogy back through `ancestral_indices` (length
    len(latents) - 1, may be empty) and returns the latents re-indexed along it
    (aesmc/inference.py:196-231)."""
    assert len(ancestral_indices) == len(latents) - 1
    probe = _first_tensor(latents[0])
    batch_size, num_particles = probe.size()[:2]
    lineage = torch.arange(num_particles, dtype=torch.int64, device=probe.device) \
        .unsqueeze(0).expand(batch_size, num_particles)
    # Systematic resampling returns non-decreasing indices; `arange` is sorted and a composition of
    # non-decreasing maps is non-decreasing, so every lineage built from K2's own outputs is sorted
    # too: the tag sends the backward of these gathers to the atomic-free segmented-sum kernel
    # (which re-checks the promise).  Indices of any other origin stay untagged.
    # ("inherited": the promise follows from the tags of other tensors, not from the kernel that wrote this one — the
    #  backward then starts from a zeroed gradient, so a false tag costs a flag and zero rows, never stale memory)
    monotone = all(getattr(index, "_aesmc_sorted", False) for index in ancestral_indices)
    if monotone:
        lineage._aesmc_sorted = "inherited"
    resampled = [None] * len(latents)
    for time in range(len(latents) - 1, -1, -1):
        resampled[time] = state.resample(latents[time], lineage)
        if time > 0:
            lineage = _ops.resample_gather(ancestral_indices[time - 1], lineage)
            if monotone:
                lineage._aesmc_sorted = "inherited"
    return resampled


def infer(inference_algorithm, observations, initial, transition, emission,
          proposal, num_particles, return_log_marginal_likelihood=False,
          return_latents=True, return_original_latents=False,
          return_log_weight=True, return_log_weights=False,
          return_ancestral_indices=False):
    """Runs 'is' or 'smc' inference (aesmc/inference.py:8-193).

    observations: length-T sequence of [batch_size, ...] tensors (or dicts of them).
    initial():                                   -> Distribution (or dict of them)
    transition(previous_latents, time, previous_observations) -> Distribution
    emission(latents, time, previous_observations)            -> Distribution
    proposal(previous_latents, time, observations)            -> Distribution with rsample
    All four are called with keyword arguments; at time 0 the proposal gets no previous_latents
    and the emission no previous_observations.

    Returns a dict with keys log_marginal_likelihood [batch_size], latents, original_latents,
    log_weight [batch_size, num_particles], log_weights, ancestral_indices (None unless requested
    through the return_* flags) and last_latent (always).

    Data-dependent failures are collected in one status word per device and raised at the end (see
    `check_device_status`); the word is shared by every stream and thread using that device, so run
    one `infer` per device at a time.  If the call is abandoned half way (an exception out of a user
    callable), whatever the kernels had flagged so far is discarded with it, so the next call starts
    clean.
    """
    try:
        begin = getattr(_kernels.get(), "begin_evaluation", None)
        if begin is not None:
            begin()
        # (`_syncfree.scope`: distributions the callables build with Python-number parameters / default validate_args
        #  neither copy from the host nor synchronise with it — the reference's own model style, test/models/lgssm.py)
        with state.deferring_draws(), _syncfree.scope():
            return _infer(inference_algorithm, observations, initial, transition, emission, proposal,
                          num_particles, return_log_marginal_likelihood, return_latents,
                          return_original_latents, return_log_weight, return_log_weights,
                          return_ancestral_indices)
    except BaseException:
        _discard_pending_flags()
        raise


def _discard_pending_flags():
    """Clears every device's status word (no synchronisation) unless a hipGraph capture is under way."""
    try:
        _kernels.get().discard_flags()
    except Exception:       # never mask the exception that is propagating
        pass


def _infer(inference_algorithm, observations, initial, transition, emission,
           proposal, num_particles, return_log_marginal_likelihood=False,
           return_latents=True, return_original_latents=False,
           return_log_weight=True, return_log_weights=False,
           return_ancestral_indices=False):
    """The body of `infer` (aesmc/inference.py:8-193).

    observations: length-T sequence of [batch_size, ...] tensors (or dicts of them).
    initial():                                   -> Distribution (or dict of them)
    transition(previous_latents, time, previous_observations) -> Distribution
    emission(latents, time, previous_observations)            -> Distribution
    proposal(previous_latents, time, observations)            -> Distribution with rsample
    All four are called with keyword arguments; at time 0 the proposal gets no previous_latents
    and the emission no previous_observations.

    Returns a dict with keys log_marginal_likelihood [batch_size], latents, original_latents,
    log_weight [batch_size, num_particles], log_weights, ancestral_indices (None unless requested
    through the return_* flags) and last_latent (always).
    """
    if inference_algorithm not in ("is", "smc"):
        raise ValueError("inference_algorithm must be either is or smc. currently = {}".format(
            inference_algorithm))
    use_smc = inference_algorithm == "smc"
    # SMC differentiates the log-weights only through their per-step row log-sum-exp unless the caller
    # asks for the weights themselves (get_loss does not): then K5 runs without an autograd node and
    # the log-sum-exp — produced by the NEXT step's resampling launch — is tied to K5's operands
    # afterwards, its backward being K5's with K1's softmax term formed in place
    fold_lse_backward = use_smc and not return_log_weight and not return_log_weights
    deferred = {}         # timestep -> what its log-sum-exp is still to be attached to: K5 / K10 operands, or a step node
    num_timesteps = len(observations)
    batch_size = _first_tensor(observations[0]).size(0)
    keep_originals = return_original_latents or return_latents

    history = []          # un-resampled draws x_0..x_t, what `emission(latents=...)` receives
    originals = []
    indices = []
    log_weights = []
    step_lse = []
    running = None        # importance sampling: sum over time of the log-weights so far (K1's accumulator)
    running_lse = None
    feed = None
    device = None
    cfg = settings.current()
    lazy_history = cfg.history_mode == "lazy"
    lazy_gather = cfg.lazy_gather
    fold_children = cfg.fold_gather_backward and use_smc and not keep_originals
    bound_rows = []       # (timestep, PendingStep): rows of the log-sum-exp stack that are values owned by step nodes

    for time in range(num_timesteps):
        if time == 0:
            ancestors = None
            proposal_dist = proposal(time=0, observations=observations)
        else:
            if use_smc:
                previous = log_weights[-1]
                if feed is None:
                    feed = _FEED_OVERRIDE.get() or _UniformFeed(batch_size, num_timesteps - 1, previous.device)
                # K2; the same launch re-indexes the newest latent (what a Markov model reads) and
                # returns the row log-sum-exp when the step before left it pending (K5 route)
                newest = history[-1] if torch.is_tensor(history[-1]) else None
                entry = deferred.get(time - 1) if step_lse[-1] is None else None
                pending = entry if isinstance(entry, _ops.PendingStep) else None    # a step node awaits this lse
                # Leave the newest latent un-gathered (`lazy_gather`) while the model's callables only describe
                # distributions in terms of it: the launch that weighs the step fetches the rows itself.  The
                # first time a step reads the values after all, the remaining steps go back to gathering inside
                # the resampling launch (one launch and one read of the indices cheaper than K2, then K3).
                lazy_step = lazy_gather and newest is not None and lazy_history and \
                    newest.dim() == 3 and newest.is_floating_point()
                # (with the latents handed to nobody, consecutive linear-Gaussian steps pass the gather's backward from
                #  node to node — `_ops.StepLink` — and the resampling launch writes the children ranges it needs)
                fold_gather = lazy_step and pending is not None and fold_children and torch.is_grad_enabled()
                index, lse_previous, moved = _ops.resample_step(previous, feed.next(), None if lazy_step else newest,
                                                                want_lse=step_lse[-1] is None, pending=pending,
                                                                want_child_end=fold_gather)
                if step_lse[-1] is None:
                    if pending is not None:
                        del deferred[time - 1]
                        if isinstance(lse_previous, _ops.LseOf):     # a value; tied to its step's node at the end
                            bound_rows.append((time - 1, lse_previous.pending))
                            lse_previous = lse_previous.value
                        step_lse[-1] = lse_previous
                    else:
                        step_lse[-1] = lse_previous if (time - 1) not in deferred else \
                            _ops.attach_lse(lse_previous, previous, deferred.pop(time - 1))
                indices.append(index)
                if lazy_history:
                    ancestors = ResampledHistory(history, index, newest=moved, lazy_newest=lazy_step)
                else:
                    ancestors = [state.resample(x, index) for x in history[:-1]]
                    ancestors.append(moved if moved is not None else state.resample(history[-1], index))
            else:
                # Importance sampling: the SAME list object that `history.append(latent)` extends below,
                # so transition(previous_latents=...) sees the current draw as previous_latents[-1] and
                # evaluates p(x_t | x_t).  Deliberate bug-compatibility: the reference aliases its list
                # the same way (`latents_bar += [latent]`, aesmc/inference.py:111, :118) and the golden
                # fixtures pin it; a model that wants p(x_t | x_{t-1}) under 'is' reads [-2].
                ancestors = history
            proposal_dist = proposal(previous_latents=ancestors, time=time,
                                     observations=observations)
        latent = state.sample(proposal_dist, batch_size, num_particles)
        history.append(latent)
        if time == 0:
            prior_dist = initial()
            emission_dist = emission(latents=history, time=0)
        else:
            prior_dist = transition(previous_latents=ancestors, time=time,
                                    previous_observations=observations[:time])
            emission_dist = emission(latents=history, time=time,
                                     previous_observations=observations[:time])
        observation = state.expand_observation(observations[time], num_particles)
        if keep_originals:
            originals.append(latent)
        # log-weight = log prior/transition + log emission - log proposal (inference.py:97-98,
        # :125-126): one kernel when all three are Normal (K5), else three summed log-densities
        # (K4 or the distribution's own log_prob) combined by K1
        log_weight_t = None
        if not isinstance(latent, dict) and not isinstance(observation, dict):
            fold = fold_lse_backward and torch.is_grad_enabled()
            log_weight_t = state.normal_log_weight(prior_dist, proposal_dist, latent, emission_dist,
                                                   observation, defer_grad=fold)
            if fold and log_weight_t is not None:
                log_weight_t, operands = log_weight_t
                if _ops.operands_require_grad(operands):
                    if getattr(operands, "is_draw", False):
                        # a linear-Gaussian step whose latent is the proposal's own draw: ONE autograd node
                        # for the step (K14).  It hands back x_t as its output — the tensor every later
                        # consumer reads; the row log-sum-exp is bound to it when a launch has produced it
                        deferred[time], x_t = _ops.affine_step(log_weight_t, operands,
                                                               fold_gather_backward=fold_children)
                        if isinstance(latent, _lazy.LazyParticles):
                            latent.resolve(x_t)     # whoever still holds the lazy draw reads this tensor
                        latent = x_t
                    else:
                        deferred[time] = operands
        # the latent as a tensor from here on: a lazy draw the fused launch formed resolves to that tensor, one that
        # no launch formed is drawn now (K9)
        latent = state.materialise_draw(latent)
        history[-1] = latent
        if keep_originals:
            originals[-1] = latent
        if lazy_gather and use_smc and time > 0 and isinstance(ancestors, ResampledHistory) and ancestors.newest_was_read():
            lazy_gather = False
        # importance sampling over several timesteps normalises the SUM of the per-step weights
        # (inference.py:156-159); K1 keeps that sum running, left to right as torch.sum over the
        # reference's stack does, and hands out its row log-sum-exp with the last step
        accumulate = (not use_smc) and time > 0 and (return_log_marginal_likelihood or return_log_weight)
        last_step = time + 1 == num_timesteps
        if log_weight_t is not None:
            # K5 gave the log-weights only; their row log-sum-exp comes out of the next resampling
            # launch for free — or from K1 when no resampling follows
            # (importance sampling never needs the per-step value: it normalises the summed weights)
            pending = use_smc and time + 1 < num_timesteps
            lse_t = None if (pending or not use_smc) else _ops.row_logsumexp(log_weight_t)
            if lse_t is not None and time in deferred:
                lse_t = _ops.attach_lse(lse_t, log_weight_t, deferred.pop(time))
            if accumulate:
                _, running, running_lse = _ops.logweight_accumulate(
                    log_weight_t, None, None, running, want_lse=last_step and return_log_marginal_likelihood)
        else:
            log_q = state.log_prob(proposal_dist, latent)
            log_p = state.log_prob(prior_dist, latent)
            log_g = state.log_prob(emission_dist, observation)
            if accumulate:
                log_weight_t, running, running_lse = _ops.logweight_accumulate(
                    log_p, log_g, log_q, running, want_lse=last_step and return_log_marginal_likelihood)
                lse_t = None
            else:
                log_weight_t, lse_t = _ops.logweight_lse(log_p, log_g, log_q)
        if not use_smc and time == 0:
            running = log_weight_t      # the first (or only) step's weights start the running sum
            if last_step:
                running_lse = lse_t     # K1 route: already at hand
        log_weights.append(log_weight_t)
        step_lse.append(lse_t)
        device = log_weight_t.device

    log_num_particles = np.log(num_particles)
    log_marginal_likelihood = latents = log_weight = None
    if use_smc:
        if return_log_marginal_likelihood:
            log_marginal_likelihood = torch.sum(
                _ops.bind_rows(torch.stack(step_lse, dim=0), bound_rows) - log_num_particles, dim=0)
        if return_latents:
            latents = get_resampled_latents(originals, indices)
        if return_log_weight:
            log_weight = log_weights[-1]
    else:
        if return_log_marginal_likelihood or return_log_weight:
            # sum over time of the per-step weights (inference.py:157): K1 kept it running, one
            # launch per step and no [T,B,K] stack; a single step is its own sum
            log_weight = running
        if return_log_marginal_likelihood:
            if running_lse is None:      # one timestep on the K5 route: no launch has reduced the rows yet
                running_lse = _ops.row_logsumexp(log_weight)
            log_marginal_likelihood = running_lse - log_num_particles
        if return_latents:
            latents = originals
        if return_original_latents:
            raise RuntimeWarning("return_original_latents shouldn't be True for is")
        if return_ancestral_indices:
            raise RuntimeWarning("return_ancestral_indices shouldn't be True for is")
        if not return_log_weight:
            log_weight = None

    capturing = device is not None and device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if device is not None and not capturing:  # under hipGraph capture the replayer checks instead
        _raise_for_flags(_kernels.get().read_flags(device))

    return {"log_marginal_likelihood": log_marginal_likelihood,
            "latents": latents,
            "original_latents": originals if (use_smc and return_original_latents) else None,
            "log_weight": log_weight,
            "log_weights": log_weights if return_log_weights else None,
            "ancestral_indices": indices if (use_smc and return_ancestral_indices) else None,
            "last_latent": latent}
