"""Optimisation loop and synthetic data source with the interface of the reference's
aesmc/train.py (train.py:10-71); `losses.get_loss` underneath runs the HIP hot path."""
import itertools
import sys

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset

from . import losses
from . import statistics


def get_chained_params(*objects):
    """One iterator over the parameters of every `nn.Module` among `objects`; None when there is no
    module at all (aesmc/train.py:10-19).  Beyond the reference, a bound method of a module (a
    model written as ONE module with `initial` / `transition` / ... methods) contributes its
    module's parameters too — once, however many of its methods are passed.  Anything else (plain
    callables, None) contributes nothing."""
    modules = []
    for candidate in objects:
        if not isinstance(candidate, nn.Module):
            candidate = getattr(candidate, "__self__", None)   # bound method -> its module
        if isinstance(candidate, nn.Module) and all(candidate is not m for m in modules):
            modules.append(candidate)
    if len(modules) == 0:
        return None
    return itertools.chain.from_iterable(m.parameters() for m in modules)


_FLAG_CHECK_INTERVAL = 32   # replays between reads of the device status word in hip_graph training


def _minibatches(dataloader, num_epochs, limit):
    """(epoch, iteration, observations) triples: every epoch walks the dataloader afresh and stops
    after `limit` minibatches when a limit is given.  As in the reference (train.py:29-32) the
    batch that trips the limit has already been fetched when the epoch ends, so a generative
    dataloader consumes its random stream identically."""
    for epoch in range(num_epochs):
        for iteration, observations in enumerate(dataloader):
            if limit is not None and iteration == limit:
                break
            yield epoch, iteration, observations


_AUTO_CAPTURE_AFTER = 8     # eager minibatches in front of the automatic capture: short runs never pay for one
_REVERIFY_EVERY = 256       # replays between checks of a replay against an eager evaluation of the same minibatch


def _default_hip_graph():
    """What `train(hip_graph=None)` means: AESMC_TRAIN_HIP_GRAPH = 'auto' (default), '0' / 'off' (the reference's eager
    loop, never a capture) or '1' / 'on' (capture the first minibatch, failures raise)."""
    import os
    choice = os.environ.get("AESMC_TRAIN_HIP_GRAPH", "auto").strip().lower()
    if choice in ("0", "off", "false", "no"):
        return False
    if choice in ("1", "on", "true", "yes"):
        return True
    return "auto"


def _capturable(observations, algorithm):
    """Host-side preconditions of a capture that can be told without trying: a sequence of HIP tensors of one device, an
    algorithm `GraphedLoss` knows, and PyTorch's own noise source in place (a test harness that replays recorded draws
    through `torch.distributions.normal._standard_normal` would have its first minibatch's draws baked into the graph)."""
    from . import state
    if algorithm not in ("iwae", "aesmc") or isinstance(observations, dict) or len(observations) == 0:
        return False
    if not all(torch.is_tensor(o) and o.is_cuda and o.device == observations[0].device for o in observations):
        return False
    return torch.distributions.normal._standard_normal is state._TORCH_STANDARD_NORMAL


def train(dataloader, num_particles, algorithm, initial, transition, emission,
          proposal, num_epochs, num_iterations_per_epoch=None,
          optimizer_algorithm=torch.optim.Adam, optimizer_kwargs={},
          callback=None, hip_graph=None, verify_replays=4, reverify_every=_REVERIFY_EVERY):
    """Fits the model parts by stochastic gradient descent on `losses.get_loss`.

    A single optimiser (`optimizer_algorithm(params, **optimizer_kwargs)`) owns the parameters of
    all four parts.  Per minibatch: clear gradients, evaluate the loss ('iwae' or 'aesmc' with
    `num_particles` particles), back-propagate, step; then, if given,
    `callback(epoch_idx, epoch_iteration_idx, loss, initial, transition, emission, proposal)`.
    Returns nothing, like the reference.

    `hip_graph` (not in the reference).  The reference's loop issues every small kernel of every timestep from Python; on
    this device that loop is host-bound below about a million particles per timestep (configs[1]: 5x).  So with 'auto'
    (what None means unless AESMC_TRAIN_HIP_GRAPH says otherwise) the loop runs eagerly for `_AUTO_CAPTURE_AFTER`
    minibatches and then — when the minibatches are HIP tensors of a fixed shape — captures loss + backward of the next
    one into one hipGraph (`graphs.GraphedLoss`) and replays it for every later minibatch of that shape.  A capture
    that cannot be made, or whose verification replays do not reproduce the eager evaluation, is abandoned with ONE
    RuntimeWarning and the loop stays eager — as it does for the whole run with `hip_graph=False` (opt-out: the
    reference's loop, step for step).  `hip_graph=True` captures the first minibatch and lets a failure raise.
    Callables in the reference's own style — `Normal(mult * x, 0.5)`, default `validate_args` — can be captured: inside
    `infer` such distributions neither copy from nor synchronise with the host (`aesmc_amd/_syncfree.py`).

    THE FROZEN-CALLABLES CONTRACT.  A replay re-issues the device work recorded at the capture; the four callables'
    Python code does not run again.  Everything they read from DEVICE memory is live (parameters, buffers and
    observations are updated in place), everything they computed on the HOST is frozen at its value during the capture:
    a Python-number coefficient annealed from `callback`, `module.train()` / `.eval()` toggles, counters, logging,
    branches on host state.  To keep that from going unnoticed, every `reverify_every`-th replay (default 256; 0 switches
    it off) is ALSO evaluated eagerly on the same minibatch with the same random draws and compared (loss and every
    gradient): on a mismatch the graph is dropped with a RuntimeWarning, that step and all later ones are the eager
    loop's.  A model that changes host-side state on purpose passes `hip_graph=False`, or keeps the state in a device
    tensor it updates in place.  A capture also holds a private memory pool beside the eager loop's.

    The capture leaves numpy's and torch's random streams where it found them and a replay consumes both exactly as an
    eager evaluation does, so a seeded run follows the eager loop's trajectory (to the rounding of identical kernels
    launched from a graph: the same bits in practice).  The fresh graph is checked against eager evaluations before it is
    used (`verify_replays` of them; 0 switches the check off).  While replaying, the device status word (NaN
    log-weights, a degenerate row, ...) is read with a synchronisation every `_FLAG_CHECK_INTERVAL` replays only; in
    between, a 4-byte asynchronous copy of the word follows every replay and is looked at when it has landed
    (`GraphedLoss.poll`: a flagged replay raises one or two replays later, without a stall), and the
    captured backward zeroes the gradients of a flagged step on the device (`GraphedLoss(guard_gradients=True)`), so the
    optimiser steps taken before the FloatingPointError / RuntimeError surfaces do not poison the parameters (with a
    stateful optimiser they still decay its moments: restore a checkpoint if that matters)."""
    model_parts = (initial, transition, emission, proposal)
    optimizer = optimizer_algorithm(get_chained_params(*model_parts), **optimizer_kwargs)
    graphed = None
    if hip_graph is None:
        hip_graph = _default_hip_graph()
    if hip_graph not in (True, False, "auto"):
        raise ValueError("aesmc_amd.train: hip_graph must be None, 'auto', True or False, got {!r}".format(hip_graph))
    may_capture = hip_graph is not False
    capture_at = 0 if hip_graph is True else _AUTO_CAPTURE_AFTER
    seen, shapes = 0, None
    for epoch, iteration, observations in _minibatches(dataloader, num_epochs, num_iterations_per_epoch):
        if may_capture and graphed is None:
            signature = None
            if _capturable(observations, algorithm):
                signature = tuple((tuple(o.shape), o.dtype, o.device) for o in observations)
            if signature is None or (shapes is not None and signature != shapes):
                if hip_graph is True:
                    raise RuntimeError("aesmc_amd: train(hip_graph=True) needs minibatches that are sequences of HIP "
                                       "tensors of one fixed shape")
                may_capture = False          # (auto: minibatches a capture cannot hold — the loop stays eager)
            shapes = signature
            if may_capture and seen >= capture_at:
                from . import graphs
                import numpy as np
                streams = (torch.cuda.get_rng_state(observations[0].device), np.random.get_state())
                try:
                    optimizer.zero_grad(set_to_none=True)   # the capture allocates the static .grad tensors
                    # check_flags=False: the device status word (NaN weights, ...) is read every
                    # _FLAG_CHECK_INTERVAL replays and once at the end instead of after each replay, so
                    # the host can prepare the next minibatch while the GPU still works on this one
                    graphed = graphs.GraphedLoss(observations, num_particles, algorithm, *model_parts,
                                                 backward=True, check_flags=False, guard_gradients=True,
                                                 verify_replays=verify_replays, preserve_random_state=True)
                except Exception as error:
                    if hip_graph is True:
                        raise
                    import warnings
                    warnings.warn("aesmc_amd.train: the loss could not be captured into a hipGraph ({}: {}); the loop "
                                  "stays eager (train(..., hip_graph=False) skips the attempt)".format(
                                      type(error).__name__, str(error)[:300]), RuntimeWarning)
                    graphed, may_capture = None, False
                    torch.cuda.synchronize()
                    # (an abandoned capture's warm-up evaluations drew random numbers: put both streams back)
                    torch.cuda.set_rng_state(streams[0], observations[0].device)
                    np.random.set_state(streams[1])
                    optimizer.zero_grad(set_to_none=True)
        seen += 1
        if graphed is not None and graphed.accepts(observations):
            if reverify_every and (graphed.replays + 1) % reverify_every == 0:
                # this step as a replay AND eagerly, same minibatch, same draws: has the callables' host side moved?
                problems, loss = _reverified_step(graphed, observations)
                if problems:
                    graphed.check()
                    graphed, may_capture = None, False      # (.grad now holds the eager evaluation's gradients)
                    optimizer.step()
                    if callback is not None:
                        callback(epoch, iteration, loss, *model_parts)
                    continue
            else:
                loss = graphed(observations)     # refreshes every captured parameter's .grad in place
            optimizer.step()
            if graphed.replays % _FLAG_CHECK_INTERVAL == 0:
                graphed.check()
            else:
                graphed.poll()       # (no synchronisation: a flagged replay surfaces a replay or two later, not up to 31)
            if callback is not None:         # the graph's loss tensor is reused by the next replay
                callback(epoch, iteration, loss.clone(), *model_parts)
            continue
        # the reference's step (aesmc/train.py:33-37); beside a live graph the static .grad tensors are kept (zeroed in place)
        optimizer.zero_grad(set_to_none=graphed is None)
        loss = losses.get_loss(observations, num_particles, algorithm, *model_parts)
        loss.backward()
        optimizer.step()
        if callback is not None:
            callback(epoch, iteration, loss, *model_parts)
        del loss      # (no eager autograd graph may be alive when a capture starts)
    if graphed is not None:
        graphed.check()


def _reverified_step(graphed, observations):
    """`graphed.reverify` with a mismatch turned into the one RuntimeWarning the caller's return to the eager loop
    deserves (an eager evaluation that does not fit beside the graph's memory pool skips the check with a warning of
    its own, `graphs.GraphedLoss._replay_against_eager`)."""
    problems, loss = graphed.reverify(observations)
    if problems:
        import warnings
        warnings.warn("aesmc_amd.train: replay {} of the captured loss no longer reproduces an eager evaluation of the "
                      "same minibatch with the same draws ({}). The callables' host-side state has changed since the "
                      "capture (a replay cannot see that); the loop goes back to eager evaluations. Pass "
                      "hip_graph=False for a model that changes such state on purpose.".format(
                          graphed.replays, "; ".join(problems)[:300]), RuntimeWarning)
    return problems, loss


class SyntheticDataset(Dataset):
    """A dataset without end: whatever index is asked for, the item is a freshly simulated
    sequence — a list of `num_timesteps` observation tensors [batch_size, ...] drawn from the
    generative model (initial, transition, emission) through `statistics.sample_from_prior`."""

    def __init__(self, initial, transition, emission, num_timesteps, batch_size):
        self.initial, self.transition, self.emission = initial, transition, emission
        self.num_timesteps, self.batch_size = num_timesteps, batch_size

    def __len__(self):
        return sys.maxsize

    def __getitem__(self, index):
        simulated = statistics.sample_from_prior(self.initial, self.transition, self.emission,
                                                 self.num_timesteps, self.batch_size)
        return [y.detach().squeeze(0) for y in simulated[1]]


def get_synthetic_dataloader(initial, transition, emission, num_timesteps, batch_size):
    """DataLoader whose every iteration yields one whole simulated batch (the loader's own batch
    size is 1 and its collate function unwraps that single item)."""
    dataset = SyntheticDataset(initial, transition, emission, num_timesteps, batch_size)
    return DataLoader(dataset, batch_size=1, collate_fn=lambda items: items[0])
