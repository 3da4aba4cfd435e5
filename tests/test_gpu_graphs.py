"""GPU tests of aesmc_amd.graphs.GraphedLoss: a whole SMC ELBO (forward, optionally backward)
replayed as one hipGraph must reproduce the eager path draw for draw."""
import numpy as np
import pytest
import torch

from aesmc_amd import graphs, losses
from aesmc_amd.testing import models

pytestmark = pytest.mark.gpu


def make(hip_device, dtype=torch.float32, d=3, B=8, T=6, affine=False):
    model = models.LgssmNd(d, seed=0, dtype=dtype, validate_args=False, affine=affine).to(hip_device)
    observations = model.simulate(T, B, seed=1)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    return model, observations, parts


def seed(value):
    torch.manual_seed(value)
    np.random.seed(value)


@pytest.mark.parametrize("algorithm,affine", [("aesmc", False), ("iwae", False), ("aesmc", True)])
def test_graphed_forward_backward_equals_eager(hip_device, algorithm, affine):
    """`affine`: AffineNormal callables — K9 / K10 / K12 (with its workspace and second launch) inside
    the captured graph."""
    model, observations, parts = make(hip_device, dtype=torch.float64, affine=affine)
    K = 64
    seed(11)
    eager_loss = losses.get_loss(observations, K, algorithm, *parts)
    eager_loss.backward()
    eager = eager_loss.detach().clone()
    del eager_loss   # a live eager autograd graph of these parameters would poison a backward capture
    eager_grads = [p.grad.clone() for p in model.parameters()]
    model.zero_grad(set_to_none=True)

    graphed = graphs.GraphedLoss(observations, K, algorithm, *parts, backward=True)
    seed(11)
    loss = graphed()
    torch.testing.assert_close(loss, eager, rtol=1e-12, atol=1e-12)
    for p, want in zip(model.parameters(), eager_grads):
        torch.testing.assert_close(p.grad, want, rtol=1e-9, atol=1e-11)

    # a replay is a fresh evaluation: new draws, gradients refreshed (not accumulated)
    first = loss.clone()
    again = graphed().clone()
    assert float((again - first).abs()) > 0
    seed(11)
    third = graphed()
    torch.testing.assert_close(third, first, rtol=1e-12, atol=1e-12)
    for p, want in zip(model.parameters(), eager_grads):
        torch.testing.assert_close(p.grad, want, rtol=1e-9, atol=1e-11)

    # new observations are copied into the static inputs
    other = model.simulate(len(observations), observations[0].size(0), seed=5)
    seed(3)
    with torch.no_grad():
        want = losses.get_loss(other, K, algorithm, *parts)
    seed(3)
    torch.testing.assert_close(graphed(other), want, rtol=1e-12, atol=1e-12)


def test_graphed_forward_only_float32(hip_device):
    model, observations, parts = make(hip_device, d=10, B=16, T=8)
    seed(2)
    with torch.no_grad():
        eager = losses.get_loss(observations, 128, "aesmc", *parts)
    graphed = graphs.GraphedLoss(observations, 128, "aesmc", *parts)
    seed(2)
    torch.testing.assert_close(graphed(), eager, rtol=1e-6, atol=1e-6)
    assert graphed.replays == 1


def test_graphed_loss_raises_deferred_errors(hip_device):
    from aesmc_amd import state
    model, observations, parts = make(hip_device)
    scale = torch.ones((), device=hip_device)

    def emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        return state.set_batch_shape_mode(
            torch.distributions.Normal(dist.loc * scale, dist.scale, validate_args=False),
            state.BatchShapeMode.FULLY_EXPANDED)

    graphed = graphs.GraphedLoss(observations, 16, "aesmc", model.initial, model.transition, emission,
                                 model.proposal)
    graphed()
    scale.fill_(float("nan"))          # poison an input the graph reads
    with pytest.raises(FloatingPointError):
        graphed()
    scale.fill_(1.0)
    assert bool(torch.isfinite(graphed()))


def test_graphed_loss_needs_hip_device():
    model = models.LgssmNd(2)
    with pytest.raises(RuntimeError):
        graphs.GraphedLoss(model.simulate(3, 2), 4, "aesmc", model.initial, model.transition,
                           model.emission, model.proposal)


def test_train_with_hip_graph_learns(hip_device):
    """train(..., hip_graph=True): the captured loss + backward, replayed per minibatch with fresh
    observations, drives the optimiser like the eager loop (loss falls; parameters move towards the
    data-generating ones)."""
    import numpy as np
    from aesmc_amd import train
    torch.manual_seed(0)
    np.random.seed(0)
    model = models.LgssmNd(2, seed=0, validate_args=False).to(hip_device)
    truth = models.LgssmNd(2, seed=1, validate_args=False).to(hip_device)
    loader = train.get_synthetic_dataloader(truth.initial, truth.transition, truth.emission, 5, 32)
    with torch.no_grad():       # start with a poor proposal (the generative parts stay stable)
        for p in (model.W0, model.b0, model.Wx, model.Wy, model.b):
            p.add_(0.5 * torch.randn_like(p))
    before = [p.detach().clone() for p in model.parameters()]
    history = []
    train.train(loader, 64, "aesmc", model.initial, model.transition, model.emission, model.proposal,
                num_epochs=2, num_iterations_per_epoch=60, optimizer_algorithm=torch.optim.Adam,
                optimizer_kwargs={"lr": 1e-2}, hip_graph=True,
                callback=lambda e, i, loss, *parts: history.append(loss.item()))
    assert len(history) == 120 and np.isfinite(history).all()
    assert np.mean(history[-10:]) < np.mean(history[:10]) - 0.5, (history[:10], history[-10:])
    assert all(not torch.equal(a, b) for a, b in zip(before, model.parameters()))


def test_train_captures_by_default_and_follows_the_eager_trajectory(hip_device):
    """train() as a user of the reference calls it — no hip_graph argument: eager for the first minibatches, then the
    captured loss + backward.  The capture consumes no random numbers and a replay consumes them as an eager step does,
    so the seeded run's losses are the eager loop's (`hip_graph=False`), minibatch for minibatch.  PyTorch's default
    `validate_args=True` is no obstacle (inside `infer` the checks stay on the device: aesmc_amd/_syncfree.py); a model
    whose callables really talk to the host (a `float(...)` of a device tensor) falls back to the eager loop with one
    warning."""
    import warnings
    import numpy as np
    from aesmc_amd import train

    class Chatty(models.LgssmNd):
        def emission(self, latents=None, time=None, previous_observations=None):
            float(self.C.sum())      # a host read of a device tensor: no capture can hold it
            return super().emission(latents=latents, time=time, previous_observations=previous_observations)

    def run(hip_graph, validate_args=False, cls=models.LgssmNd):
        torch.manual_seed(3)
        np.random.seed(3)
        model = cls(3, seed=0, validate_args=validate_args).to(hip_device)
        truth = models.LgssmNd(3, seed=1, validate_args=False).to(hip_device)
        loader = train.get_synthetic_dataloader(truth.initial, truth.transition, truth.emission, 6, 16)
        history = []
        train.train(loader, 128, "aesmc", model.initial, model.transition, model.emission, model.proposal,
                    num_epochs=1, num_iterations_per_epoch=train._AUTO_CAPTURE_AFTER + 12,
                    optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 1e-3}, hip_graph=hip_graph,
                    callback=lambda e, i, loss, *parts: history.append(loss.item()))
        return np.asarray(history), [p.detach().clone() for p in model.parameters()]
    eager, eager_params = run(False)
    calls = {"replays": 0}
    from aesmc_amd import graphs
    real = graphs.GraphedLoss.__call__

    def counting(self, *args, **kwargs):
        calls["replays"] += 1
        return real(self, *args, **kwargs)
    graphs.GraphedLoss.__call__ = counting
    try:
        auto, auto_params = run(None)
    finally:
        graphs.GraphedLoss.__call__ = real
    assert calls["replays"] == 12, calls            # the minibatches behind the eager ones were replays
    assert len(auto) == len(eager) == train._AUTO_CAPTURE_AFTER + 12
    np.testing.assert_array_equal(auto[:train._AUTO_CAPTURE_AFTER], eager[:train._AUTO_CAPTURE_AFTER])
    np.testing.assert_allclose(auto, eager, rtol=2e-5)
    for a, b in zip(auto_params, eager_params):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    # PyTorch's default validate_args: captured all the same, no warning, the eager loop's numbers
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        calls["replays"] = 0
        graphs.GraphedLoss.__call__ = counting
        try:
            validated, _ = run(None, validate_args=None)
        finally:
            graphs.GraphedLoss.__call__ = real
    assert calls["replays"] == 12 and not [w for w in caught if "captured" in str(w.message)], [str(w.message) for w in caught]
    want, _ = run(False, validate_args=None)
    np.testing.assert_allclose(validated, want, rtol=2e-5)
    # not capturable: one warning, the eager loop's numbers
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        fallback, _ = run(None, cls=Chatty)
    want, _ = run(False, cls=Chatty)
    assert sum("could not be captured" in str(w.message) for w in caught) == 1
    np.testing.assert_allclose(fallback, want, rtol=2e-5)


def test_graphs_captured_into_recycled_memory_give_eager_gradients(hip_device):
    """Regression: a graph captured after an earlier one was destroyed gets the earlier graph's
    memory back, dirty.  Every gradient of a replay must still equal the eager one bit for bit —
    i.e. nothing in the captured backward may depend on what a recycled block held (K3's backward
    zero-fills rows without offspring; a captured hipMemsetAsync node was seen to run too early)."""
    import gc
    dtype = torch.float32
    for round_ in range(3):
        seed(0)
        model = models.LgssmNd(3, seed=0, dtype=dtype, validate_args=False).to(hip_device)
        parts = (model.initial, model.transition, model.emission, model.proposal)
        observations = model.simulate(6, 8, seed=3)
        graphed = graphs.GraphedLoss(observations, 64, "aesmc", *parts, backward=True)
        params = list(model.parameters())
        seed(100 + round_)
        graph_loss = graphed(observations).clone()
        graph_grads = [p.grad.clone() for p in params]
        static = [p.grad for p in params]
        for p in params:
            p.grad = None
        seed(100 + round_)
        eager_loss = losses.get_loss(observations, 64, "aesmc", *parts)
        eager_loss.backward()
        assert torch.equal(graph_loss, eager_loss.detach())
        for name, got, p in zip([n for n, _ in model.named_parameters()], graph_grads, params):
            assert torch.equal(got, p.grad), (round_, name)
        for p, grad in zip(params, static):
            p.grad = grad
        del eager_loss, graphed, model, parts, params, static
        gc.collect()


def test_replayed_gradients_stay_equal_to_eager_over_a_training_run(hip_device):
    """Regression for ROCm 7.0's hipGraph fast path running captured MEMSET nodes out of stream
    order (PyTorch's multi-block reductions zero their semaphores that way): at configs[1]-like
    sizes the replayed gradients of the small broadcast parameters went wrong from the fourth
    iteration on while the loss stayed right.  aesmc_amd switches the fast path off on import
    (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0); every replay — at whatever state the random streams are in,
    with eager work and optimiser steps in between — must equal the eager gradients bit for bit."""
    import gc
    import os
    import aesmc_amd
    assert os.environ.get(aesmc_amd.HIPGRAPH_ENV) == "0"
    dim, K, B, T = 10, 1024, 256, 20
    seed(0)
    truth = models.LgssmNd(dim, seed=1, validate_args=False).to(hip_device)
    model = models.LgssmNd(dim, seed=0, validate_args=False).to(hip_device)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]
    optimizer = torch.optim.Adam(params, lr=3e-3)
    graphed = None
    for iteration in range(8):
        observations = truth.simulate(T, B, seed=50 + iteration)
        torch.randn(1000, device=hip_device)                 # eager use of the generator in between
        if graphed is None:
            graphed = graphs.GraphedLoss(observations, K, "aesmc", *parts, backward=True)
        cuda_state, numpy_state = torch.cuda.get_rng_state(hip_device), np.random.get_state()
        graph_loss = graphed(observations).clone()
        graph_grads = [p.grad.clone() for p in params]
        static = [p.grad for p in params]
        for p in params:
            p.grad = None
        torch.cuda.set_rng_state(cuda_state, hip_device)
        np.random.set_state(numpy_state)
        eager_loss = losses.get_loss(observations, K, "aesmc", *parts)
        eager_loss.backward()
        assert torch.equal(graph_loss, eager_loss.detach()), iteration
        for name, got, p in zip(names, graph_grads, params):
            assert torch.equal(got, p.grad), (iteration, name, float((got - p.grad).abs().max()))
        for p, grad in zip(params, static):
            p.grad = grad
        del eager_loss
        gc.collect()
        optimizer.step()


def test_capture_failure_explains_itself(hip_device):
    """A callable that talks to the host (a scale uploaded with `torch.tensor(..., device=)` on every call — a
    pageable host-to-device copy) cannot be captured; the error must say what to change.  (A Python-number parameter
    as such is fine: inside `infer` it is a cached device constant, tests/test_gpu_reference_models.py.)"""
    from aesmc_amd import state
    model, observations, parts = make(hip_device)

    def emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        scale = torch.tensor(0.5, device=hip_device)
        return state.set_batch_shape_mode(torch.distributions.Normal(dist.loc, scale, validate_args=False),
                                          state.BatchShapeMode.FULLY_EXPANDED)

    with pytest.raises(RuntimeError, match="could not be captured"):
        graphs.GraphedLoss(observations, 16, "aesmc", model.initial, model.transition, emission, model.proposal)
    torch.cuda.synchronize()
    # the device and the package are still usable afterwards
    assert bool(torch.isfinite(losses.get_loss(observations, 16, "aesmc", *parts)))


def test_distributed_train_on_a_one_rank_rccl_group_equals_single_process_training(hip_device):
    """distributed.train (eager and hip_graph) with a one-rank RCCL group takes the sharded code
    path — shard scope, all-reduce of the loss and of the flat gradient bucket — and must land on
    exactly the parameters train.train reaches from the same seeds.  Every second replay is re-verified against an eager
    evaluation (`reverify_every=2`: the ranks' agreed verdict, the random streams moved by exactly one evaluation): the
    trajectory must not notice."""
    import torch.distributed as dist
    from aesmc_amd import distributed, train
    created = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as probe:
            probe.bind(("127.0.0.1", 0))
            port = probe.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{}".format(port), rank=0, world_size=1,
                                device_id=hip_device)
        created = True
    try:
        results = {}
        for label, fn, kwargs in (("single", train.train, {}), ("sharded", distributed.train, {}),
                                  ("single_graph", train.train, {"hip_graph": True, "reverify_every": 2}),
                                  ("sharded_graph", distributed.train, {"hip_graph": True, "reverify_every": 2})):
            seed(0)
            truth = models.LgssmNd(2, seed=1, validate_args=False).to(hip_device)
            model = models.LgssmNd(2, seed=0, validate_args=False).to(hip_device)
            loader = [truth.simulate(4, 16, seed=20 + i) for i in range(6)]
            seen = []
            fn(loader, 32, "aesmc", model.initial, model.transition, model.emission, model.proposal,
               num_epochs=1, num_iterations_per_epoch=5, optimizer_algorithm=torch.optim.SGD,
               optimizer_kwargs={"lr": 1e-2}, callback=lambda e, i, loss, *parts: seen.append(loss.item()),
               **kwargs)
            results[label] = (seen, [p.detach().clone() for p in model.parameters()])
        for a, b in (("single", "sharded"), ("single_graph", "sharded_graph")):
            np.testing.assert_allclose(results[a][0], results[b][0], rtol=1e-6)
            for pa, pb in zip(results[a][1], results[b][1]):
                torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-7)
    finally:
        if created:
            dist.destroy_process_group()


def test_backward_capture_refuses_an_unprotected_runtime(hip_device):
    """the hipGraph memset workaround must be in effect when the HIP runtime
    starts.  In fresh processes: (a) the variable set to 1 -> GraphedLoss(backward=True) raises instead
    of warning; (b) the runtime started before the import (torch.cuda.is_available() does that without
    setting torch.cuda.is_initialized()) -> the package notices ('too-late') and the capture raises;
    (c) imported first -> 'set', and the capture verifies itself against eager."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = """
import sys, torch
sys.path.insert(0, {root!r})
{before}
import aesmc_amd
from aesmc_amd import graphs
from aesmc_amd.testing import models
print("STATUS", aesmc_amd.HIPGRAPH_MEMSET_WORKAROUND)
dev = torch.device("cuda", 0)
model = models.LgssmNd(2, seed=0, validate_args=False).to(dev)
obs = model.simulate(3, 4, seed=0)
try:
    graphs.GraphedLoss(obs, 16, "aesmc", model.initial, model.transition, model.emission, model.proposal, backward=True)
    print("CAPTURED")
except RuntimeError as error:
    print("REFUSED", str(error)[:80])
"""
    def run(before, env_value):
        env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
        if env_value is not None:
            env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = env_value
        done = subprocess.run([sys.executable, "-c", body.format(root=root, before=before)], env=env,
                              capture_output=True, text=True, timeout=600)
        assert done.returncode == 0, done.stderr[-2000:]
        return done.stdout
    out = run("", "1")
    assert "STATUS preset:1" in out and "REFUSED" in out
    out = run("torch.cuda.is_available()", None)
    assert "STATUS too-late" in out and "REFUSED" in out
    out = run("", None)
    assert "STATUS set" in out and "CAPTURED" in out


def test_guarded_gradients_are_zero_on_a_flagged_replay(hip_device):
    """`GraphedLoss(guard_gradients=True)` (what train(hip_graph=True) captures): a replay whose
    kernels flag NaN log-weights leaves every gradient exactly zero — an optimiser step taken before
    the host reads the status word cannot poison the parameters — and the check then raises; a
    healthy replay's gradients equal the unguarded ones bit for bit."""
    seed(0)
    model = models.LgssmNd(3, seed=0, validate_args=False).to(hip_device)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    observations = model.simulate(5, 8, seed=3)
    params = list(model.parameters())
    plain = graphs.GraphedLoss(observations, 64, "aesmc", *parts, backward=True)
    seed(7)
    plain(observations)
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    del plain
    guarded = graphs.GraphedLoss(observations, 64, "aesmc", *parts, backward=True, guard_gradients=True,
                                 check_flags=False)
    seed(7)
    guarded(observations)
    guarded.check()
    for p, g in zip(params, want):
        assert torch.equal(p.grad, g)
    poisoned = [o.clone() for o in observations]
    poisoned[2][1, 0] = float("nan")
    guarded(poisoned)
    assert all(bool((p.grad == 0).all()) for p in params)
    with pytest.raises(FloatingPointError):
        guarded.check()
    guarded(observations)                         # and the next healthy minibatch trains on
    guarded.check()
    assert any(bool((p.grad != 0).any()) for p in params)


@pytest.mark.gpu
def test_a_replay_draws_its_uniforms_one_timestep_at_a_time(hip_device):
    """The uniforms of a replay are drawn as T - 1 calls of `np.random.uniform(size=[B, 1])` — the reference's own
    consumption of the global RandomState (aesmc/inference.py:250), the numbers the eager loop draws — and never as
    one [T-1, B, 1] array: a host allocation of hundreds of KB made and freed before every replay stalled the device
    for tens of milliseconds on the MI355X stack (profiles/README.md, tools/graph_probe.py)."""
    T, B, K = 12, 8, 64
    model = models.LgssmNd(3, dtype=torch.float32, affine=True, validate_args=False).tune_proposal().to(hip_device)
    observations = model.simulate(T, B, seed=1)
    np.random.seed(3)
    torch.manual_seed(3)
    graphed = graphs.GraphedLoss(observations, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
    sizes = []
    real = np.random.uniform

    def spy(*args, **kwargs):
        sizes.append(tuple(kwargs.get("size", args[2] if len(args) > 2 else ())))
        return real(*args, **kwargs)

    np.random.uniform = spy
    try:
        np.random.seed(5)
        torch.manual_seed(5)
        replayed = float(graphed())
    finally:
        np.random.uniform = real
    assert sizes == [(B, 1)] * (T - 1)
    np.random.seed(5)
    torch.manual_seed(5)
    eager = float(losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission, model.proposal))
    assert abs(replayed - eager) <= 1e-5 * max(1.0, abs(eager))


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["aesmc", "iwae"])
def test_back_to_back_replays_each_read_their_own_generator_offset(hip_device, algorithm):
    """Replays enqueued without a host sync in between (check_flags=False: what `train(hip_graph=True)` and
    `distributed.train` do) must each see the generator offset of THEIR evaluation: the pinned word the upload
    copies from is one of a ring of slots, none rewritten before its copy has completed.  The state the device held
    during each replay is cloned on the stream right after it and compared with the offsets the generator had
    before each replay.  ('iwae' has no uniform feed whose event would hold the host back at all.)"""
    from aesmc_amd import _philox, state
    T, B, K = (6, 8, 64) if algorithm == "aesmc" else (1, 16, 256)
    if algorithm == "aesmc":
        model = models.LgssmNd(3, dtype=torch.float32, affine=True, validate_args=False).tune_proposal().to(hip_device)
        parts = (model.initial, model.transition, model.emission, model.proposal)
    else:
        model = models.GaussianIwae(state=state, validate_args=False).to(hip_device)
        parts = (model.initial, None, model.emission, model.proposal)
    observations = model.simulate(T, B, seed=1)
    np.random.seed(3)
    torch.manual_seed(3)
    graphed = graphs.GraphedLoss(observations, K, algorithm, *parts, check_flags=False)
    if graphed.noise is None:
        pytest.skip("this model's draws go through PyTorch's own captured generator state")
    generator = torch.cuda.default_generators[hip_device.index]
    busy = torch.empty(64 << 20, device=hip_device)
    before, seen = [], []
    for _ in range(3 * _philox.GraphNoise.SLOTS):
        busy.normal_()                       # device work in front: the host gets well ahead of the replays
        before.append(generator.get_offset())
        graphed()
        seen.append(graphed.noise.state.clone())
    torch.cuda.synchronize()
    assert [int(s[1]) for s in seen] == before
    assert len(set(before)) == len(before)          # and every replay advanced the generator


@pytest.mark.parametrize("kind", ["wide24", "wide192", "nonlinear_fused"])
def test_round6_kernels_inside_a_captured_loss_and_backward(hip_device, kind):
    """The kernels added in round 6 inside `GraphedLoss(backward=True)`: K17g / K18g (rows of 24 and of 192 values — the
    chunked form — with their recomputing backward and its workspace allocations) and K13 / K13b (the proposal net of the
    nonlinear model, its records and the binder's sums).  The capture verifies itself (4 replays against eager
    evaluations on the same draws); a seeded replay afterwards equals the seeded eager evaluation."""
    if kind == "nonlinear_fused":
        model = models.NonlinearSsm(6, hidden=32, seed=0, dtype=torch.float32, validate_args=False, fused=True).to(hip_device)
        B, K, T = 4, 256, 4
    else:
        dim = 24 if kind == "wide24" else 192
        model = models.LgssmNd(dim, seed=0, dtype=torch.float32, validate_args=False, affine=True,
                               emission_scale=0.5).tune_proposal().to(hip_device)
        B, K, T = 3, 96, 4
    observations = model.simulate(T, B, seed=1)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    seed(13)
    eager_loss = losses.get_loss(observations, K, "aesmc", *parts)
    eager_loss.backward()
    eager = eager_loss.detach().clone()
    del eager_loss
    eager_grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    graphed = graphs.GraphedLoss(observations, K, "aesmc", *parts, backward=True, verify_replays=4)
    seed(13)
    loss = graphed()
    torch.testing.assert_close(loss, eager, rtol=2e-6, atol=2e-6)
    for name, p in model.named_parameters():
        if name in eager_grads:
            want = eager_grads[name]
            scale = float(want.abs().max()) + 1e-30
            assert float((p.grad - want).abs().max()) <= 1e-4 * scale, name


def test_a_flagged_replay_surfaces_within_a_few_replays_without_a_stall(hip_device):
    """the captured training loop reads the device status word with a synchronisation every 32
    replays only — a NaN minibatch used to surface up to 31 optimiser steps late.  `GraphedLoss.poll` looks at an
    asynchronous 4-byte copy of the word after every replay: the FloatingPointError is raised within three replays of the
    one that was flagged (and the flagged steps' gradients were zeroed on the device meanwhile)."""
    from aesmc_amd import train
    seed(0)
    truth = models.LgssmNd(2, seed=1, validate_args=False).to(hip_device)
    model = models.LgssmNd(2, seed=0, validate_args=False).to(hip_device)
    loader = [truth.simulate(4, 16, seed=20 + i) for i in range(40)]
    seen = []

    def callback(epoch, iteration, loss, *parts):
        seen.append(iteration)
        if iteration == 4:
            with torch.no_grad():
                model.A.fill_(float("nan"))      # every later replay's log-weights are NaN
    with pytest.raises(FloatingPointError):
        train.train(loader, 32, "aesmc", model.initial, model.transition, model.emission, model.proposal, num_epochs=1,
                    optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 1e-3}, callback=callback, hip_graph=True,
                    reverify_every=0)
    assert 4 < seen[-1] <= 8, seen      # (iteration 5 is the first flagged replay)
    from aesmc_amd import inference
    inference.check_device_status(hip_device)      # the word was cleared when it was raised
