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


def train(dataloader, num_particles, algorithm, initial, transition, emission,
          proposal, num_epochs, num_iterations_per_epoch=None,
          optimizer_algorithm=torch.optim.Adam, optimizer_kwargs={},
          callback=None, hip_graph=False, verify_replays=4):
    """Fits the model parts by stochastic gradient descent on `losses.get_loss`.

    A single optimiser (`optimizer_algorithm(params, **optimizer_kwargs)`) owns the parameters of
    all four parts.  Per minibatch: clear gradients, evaluate the loss ('iwae' or 'aesmc' with
    `num_particles` particles), back-propagate, step; then, if given,
    `callback(epoch_idx, epoch_iteration_idx, loss, initial, transition, emission, proposal)`.
    Returns nothing, like the reference.

    `hip_graph=True` (not in the reference; off by default) captures loss + backward of the first
    minibatch into one hipGraph (`graphs.GraphedLoss`) and replays it for every later one — the
    loop is then no longer bound by the host issuing each small kernel.  It needs what any capture
    needs (fixed minibatch shapes, tensor observations, callables that never synchronise with the
    host; see `aesmc_amd/graphs.py`) and its warm-up evaluations consume random numbers, so a
    seeded run follows a different — equally distributed — trajectory than the eager loop.
    The fresh graph is checked against eager evaluations before it is used (`verify_replays` of them, each a full
    eager forward + backward with its memory peak; 0 switches the check off — see `graphs.GraphedLoss`).
    The device status word (NaN log-weights, a degenerate row, ...) is then read every
    `_FLAG_CHECK_INTERVAL` replays instead of every step; in between, the captured backward zeroes the
    gradients of a flagged step on the device (`GraphedLoss(guard_gradients=True)`), so the optimiser
    steps taken before the FloatingPointError / RuntimeError surfaces do not poison the parameters
    (with a stateful optimiser they still decay its moments: restore a checkpoint if that matters)."""
    model_parts = (initial, transition, emission, proposal)
    optimizer = optimizer_algorithm(get_chained_params(*model_parts), **optimizer_kwargs)
    if hip_graph:
        from . import graphs
        graphed = None
        for epoch, iteration, observations in _minibatches(dataloader, num_epochs, num_iterations_per_epoch):
            if graphed is None:
                optimizer.zero_grad(set_to_none=True)   # the capture allocates the static .grad tensors
                # check_flags=False: the device status word (NaN weights, ...) is read every
                # _FLAG_CHECK_INTERVAL replays and once at the end instead of after each replay, so
                # the host can prepare the next minibatch while the GPU still works on this one
                graphed = graphs.GraphedLoss(observations, num_particles, algorithm, *model_parts,
                                             backward=True, check_flags=False, guard_gradients=True,
                                             verify_replays=verify_replays)
            loss = graphed(observations)     # refreshes every captured parameter's .grad in place
            optimizer.step()
            if graphed.replays % _FLAG_CHECK_INTERVAL == 0:
                graphed.check()
            if callback is not None:         # the graph's loss tensor is reused by the next replay
                callback(epoch, iteration, loss.clone(), *model_parts)
        if graphed is not None:
            graphed.check()
        return
    for epoch, iteration, observations in _minibatches(dataloader, num_epochs, num_iterations_per_epoch):
        optimizer.zero_grad()
        loss = losses.get_loss(observations, num_particles, algorithm, *model_parts)
        loss.backward()
        optimizer.step()
        if callback is not None:
            callback(epoch, iteration, loss, *model_parts)


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
