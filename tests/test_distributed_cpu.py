"""Batch sharding across ranks on CPU: two gloo processes, kernels
substituted by the oracle.  A sharded ELBO must equal the unsharded one: same ancestor indices row
for row (every rank consumes the full uniform block and keeps its rows), loss equal after ONE
all-reduce of sum log Z, parameter gradients equal after the flat-bucket all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

B, K, T, D = 6, 32, 4, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tape_for(model, observations):
    """Noise blocks for the whole batch so every rank can replay its own rows."""
    from aesmc_amd.testing import replay
    gen = torch.Generator().manual_seed(123)
    normals = [torch.randn(K, B, D, generator=gen, dtype=torch.float64).numpy()]
    normals += [torch.randn(B, K, D, generator=gen, dtype=torch.float64).numpy() for _ in range(T - 1)]
    uniforms = [np.random.RandomState(7 + t).uniform(size=(B, 1)) for t in range(T - 1)]
    return replay.Tape(normals, uniforms)


def _slice_tape(tape, lo, hi):
    from aesmc_amd.testing import replay
    normals = [tape.normals[0][:, lo:hi]] + [n[lo:hi] for n in tape.normals[1:]]
    return replay.Tape(normals, tape.uniforms)  # uniforms stay global: shard_scope slices them


def _run_rank(rank, world, port, out_dir, affine=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aesmc_amd import _kernels, distributed, inference
        from aesmc_amd.testing import models, replay
        from tests.oracle_provider import OracleKernels
        _kernels._swap_provider_for_tests(OracleKernels())
        torch.set_num_threads(1)
        model = models.LgssmNd(D, seed=0, dtype=torch.float64, affine=affine)
        observations = model.simulate(T, B, seed=1)
        tape = _tape_for(model, observations)
        lo, hi = distributed.shard_bounds(B, rank, world)
        local_obs = distributed.shard_observations(observations, rank, world)
        assert local_obs[0].shape[0] == hi - lo
        parts = (model.initial, model.transition, model.emission, model.proposal)
        with replay.replay(_slice_tape(tape, lo, hi)):
            loss = distributed.sharded_get_loss(local_obs, K, "aesmc", *parts, global_batch_size=B)
        loss.backward()
        distributed.all_reduce_gradients(list(model.parameters()))
        with replay.replay(_slice_tape(tape, lo, hi)):
            exact = distributed.sharded_get_loss(local_obs, K, "aesmc", *parts, global_batch_size=B,
                                                 exact_mean=True)
        exact_grads = torch.autograd.grad(exact, list(model.parameters()))
        exact_grads = [g.clone() for g in exact_grads]
        for g in exact_grads:
            dist.all_reduce(g)
        with replay.replay(_slice_tape(tape, lo, hi)), distributed.shard_scope(B, rank, world):
            result = inference.infer("smc", local_obs, *parts, K, return_ancestral_indices=True,
                                     return_latents=False)
        # importance sampling over T steps: the running sum of K1 instead of a [T,B,K] stack
        with replay.replay(_slice_tape(tape, lo, hi)):
            iwae = distributed.sharded_get_loss(local_obs, K, "iwae", *parts, global_batch_size=B)
        iwae_grads = [g.clone() for g in torch.autograd.grad(iwae, list(model.parameters()), allow_unused=True)
                      if g is not None]
        for g in iwae_grads:
            dist.all_reduce(g)
        torch.save({"loss": loss.detach(), "grads": [p.grad.clone() for p in model.parameters()],
                    "iwae_loss": iwae.detach(), "iwae_grads": iwae_grads,
                    "exact_loss": exact.detach(), "exact_grads": exact_grads,
                    "indices": result["ancestral_indices"], "rows": (lo, hi)},
                   os.path.join(out_dir, "rank{}.pt".format(rank)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("affine", [False, True])
def test_two_rank_batch_shard_matches_single_process(tmp_path, oracle_backend, affine):
    """`affine`: the model's callables return AffineNormal — the sharded run goes through the
    linear-Gaussian route (K9 / K10 / K12 on the oracle backend) on every rank."""
    world = 2
    port = _free_port()
    mp.spawn(_run_rank, args=(world, port, str(tmp_path), affine), nprocs=world, join=True)

    from aesmc_amd import inference, losses
    from aesmc_amd.testing import models, replay
    model = models.LgssmNd(D, seed=0, dtype=torch.float64, affine=affine)
    observations = model.simulate(T, B, seed=1)
    tape = _tape_for(model, observations)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    with replay.replay(tape):
        loss = losses.get_loss(observations, K, "aesmc", *parts)
    loss.backward()
    with replay.replay(tape):
        full = inference.infer("smc", observations, *parts, K, return_ancestral_indices=True,
                               return_latents=False)

    with replay.replay(tape):
        iwae = losses.get_loss(observations, K, "iwae", *parts)
    iwae_grads = [g for g in torch.autograd.grad(iwae, list(model.parameters()), allow_unused=True) if g is not None]

    shards = [torch.load(os.path.join(str(tmp_path), "rank{}.pt".format(r))) for r in range(world)]
    assert shards[0]["rows"] == (0, 3) and shards[1]["rows"] == (3, 6)
    for shard in shards:
        torch.testing.assert_close(shard["iwae_loss"], iwae.detach(), rtol=1e-12, atol=1e-12)
        assert len(shard["iwae_grads"]) == len(iwae_grads) > 0
        for got, want in zip(shard["iwae_grads"], iwae_grads):
            torch.testing.assert_close(got, want, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(shard["loss"], loss.detach(), rtol=1e-12, atol=1e-12)
        assert torch.equal(shard["exact_loss"], loss.detach())       # all-gather + torch.mean: to the last bit
        for got, want in zip(shard["exact_grads"], [p.grad for p in model.parameters()]):
            torch.testing.assert_close(got, want, rtol=1e-10, atol=1e-12)
        lo, hi = shard["rows"]
        for got, want in zip(shard["indices"], full["ancestral_indices"]):
            assert torch.equal(got, want[lo:hi])
        for got, want in zip(shard["grads"], [p.grad for p in model.parameters()]):
            torch.testing.assert_close(got, want, rtol=1e-10, atol=1e-12)


def _train_rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aesmc_amd import _kernels, distributed, train
        from aesmc_amd.testing import models
        from tests.oracle_provider import OracleKernels
        _kernels._swap_provider_for_tests(OracleKernels())
        torch.set_num_threads(1)
        np.random.seed(5)                  # same on every rank: global uniform blocks
        torch.manual_seed(100 + rank)      # different per rank: own data rows, own proposal noise
        truth = models.LgssmNd(2, seed=1, dtype=torch.float64)
        model = models.LgssmNd(2, seed=0, dtype=torch.float64)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(0.5)
        loader = train.get_synthetic_dataloader(truth.initial, truth.transition, truth.emission, 3, 4)
        losses_seen = []
        distributed.train(loader, 16, "aesmc", model.initial, model.transition, model.emission,
                          model.proposal, num_epochs=1, num_iterations_per_epoch=6,
                          optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.01},
                          callback=lambda e, i, loss, *parts: losses_seen.append(loss.item()))
        torch.save({"params": [p.detach().clone() for p in model.parameters()], "losses": losses_seen},
                   os.path.join(out_dir, "train_rank{}.pt".format(rank)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_training_keeps_replicas_identical(tmp_path, oracle_backend):
    """distributed.train: own data rows and proposal noise per rank, one all-reduce for the loss and
    one for the gradients per step -> both ranks report the same global loss and hold the same
    parameters after every step, and those parameters moved."""
    world = 2
    mp.spawn(_train_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = [torch.load(os.path.join(str(tmp_path), "train_rank{}.pt".format(r))) for r in range(world)]
    assert len(a["losses"]) == 6 and np.isfinite(a["losses"]).all()
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-12)
    from aesmc_amd.testing import models
    start = models.LgssmNd(2, seed=0, dtype=torch.float64)
    for pa, pb, p0 in zip(a["params"], b["params"], start.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-12, atol=1e-14)
        assert not torch.allclose(pa, 0.5 * p0.detach())


def test_shard_bounds_cover_the_batch():
    from aesmc_amd import distributed
    for batch in (1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    obs = [{"y": torch.arange(8.0)}, {"y": torch.arange(8.0)}]
    cut = distributed.shard_observations(obs, 1, 2)
    assert torch.equal(cut[0]["y"], torch.arange(4.0, 8.0))
