"""Optimisation loop and synthetic data source with the interface of the reference's
aesmc/train.py; `losses.get_loss` underneath runs the HIP hot path."""
import itertools
import sys

import torch
import torch.nn as nn
import torch.utils.data

from . import losses
from . import statistics


def get_chained_params(*objects):
    """Parameters of every nn.Module among `objects`, chained; None if there is none
    (aesmc/train.py:10-19)."""
    modules = [obj for obj in objects if isinstance(obj, nn.Module)]
    if not modules:
        return None
    return itertools.chain.from_iterable(module.parameters() for module in modules)


def train(dataloader, num_particles, algorithm, initial, transition, emission,
          proposal, num_epochs, num_iterations_per_epoch=None,
          optimizer_algorithm=torch.optim.Adam, optimizer_kwargs={},
          callback=None):
    """One optimiser over the parameters of the four model parts; per minibatch: zero_grad,
    get_loss, backward, step, then callback(epoch_idx, epoch_iteration_idx, loss, initial,
    transition, emission, proposal) (aesmc/train.py:22-41)."""
    optimizer = optimizer_algorithm(
        get_chained_params(initial, transition, emission, proposal), **optimizer_kwargs)
    for epoch_idx in range(num_epochs):
        for epoch_iteration_idx, observations in enumerate(dataloader):
            if num_iterations_per_epoch is not None and \
                    epoch_iteration_idx == num_iterations_per_epoch:
                break
            optimizer.zero_grad()
            loss = losses.get_loss(observations, num_particles, algorithm, initial, transition,
                                   emission, proposal)
            loss.backward()
            optimizer.step()
            if callback is not None:
                callback(epoch_idx, epoch_iteration_idx, loss, initial, transition, emission,
                         proposal)


class SyntheticDataset(torch.utils.data.Dataset):
    """Endless dataset: every item is a fresh length-num_timesteps list of [batch_size, ...]
    observations drawn from the generative model (aesmc/train.py:44-62)."""

    def __init__(self, initial, transition, emission, num_timesteps, batch_size):
        self.initial = initial
        self.transition = transition
        self.emission = emission
        self.num_timesteps = num_timesteps
        self.batch_size = batch_size

    def __getitem__(self, index):
        _, observations = statistics.sample_from_prior(
            self.initial, self.transition, self.emission, self.num_timesteps, self.batch_size)
        return [observation.detach().squeeze(0) for observation in observations]

    def __len__(self):
        return sys.maxsize


def get_synthetic_dataloader(initial, transition, emission, num_timesteps, batch_size):
    """DataLoader over SyntheticDataset that yields one whole batch per iteration
    (aesmc/train.py:65-71)."""
    return torch.utils.data.DataLoader(
        SyntheticDataset(initial, transition, emission, num_timesteps, batch_size),
        batch_size=1, collate_fn=lambda items: items[0])
