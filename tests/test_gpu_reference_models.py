"""The reference's OWN model style on the device's fast lanes.

test/models/lgssm.py:8-72 and test/models/gaussian.py:6-47 — restated literally in `aesmc_amd.testing.models`
(Lgssm1d*, Gaussian*: Python-number scales, PyTorch's default `validate_args`, the `cat` / `view` proposal, a
host-resident prior `std`) — must (a) run through `infer` without a host-to-device copy or a host read per
distribution, (b) be capturable into a hipGraph whose replays equal the eager evaluation, (c) train through
`train.train`'s captured loop along the eager loop's trajectory, and (d) reproduce the reference's own recorded training
runs (tests/golden/train_*.npz) THROUGH the captured loop.
"""
import warnings

import numpy as np
import pytest
import torch

from aesmc_amd import _syncfree, graphs, inference, losses, state, train
from aesmc_amd.testing import models, replay
from tests.golden_io import Golden
from tests.test_gpu_configs_parity import TRAIN_CASES, _train_case_parts

pytestmark = pytest.mark.gpu


def lgssm1d(device, transition_mult=0.7, emission_mult=0.9):
    """The four parts as test/test_inference.py / test_losses.py build them (floats everywhere), moved to the device."""
    parts = (models.Lgssm1dInitial(0.0, 1.0),
             models.Lgssm1dTransition(transition_mult, 0.5).to(device),
             models.Lgssm1dEmission(emission_mult, 0.4).to(device),
             models.Lgssm1dProposal(0.8, 0.8).to(device))
    return parts


def gaussian(device):
    return (models.GaussianPrior(0.3, 1.0).to(device), None, models.GaussianLikelihood(0.8).to(device),
            models.GaussianInferenceNetwork(0.5, 0.1, 0.9).to(device))


def observations_for(family, T, B, device, seed=3):
    """Data from the model itself, generated on the host as the reference's runs do (its `Initial` is a host `Normal`)."""
    torch.manual_seed(seed)
    cpu = torch.device("cpu")
    true = lgssm1d(cpu) if family == "lgssm1d" else gaussian(cpu)
    loader = train.get_synthetic_dataloader(true[0], true[1], true[2], T, B)
    return [o.to(device) for o in next(iter(loader))]


def seed(value):
    torch.manual_seed(value)
    np.random.seed(value)


class _CopySpy:
    """Counts host <-> device copies and host reads of device tensors made through the Python API."""

    def __init__(self):
        self.events = []

    def __enter__(self):
        self._to, self._item, self._tensor = torch.Tensor.to, torch.Tensor.item, torch.tensor
        self._all = torch._is_all_true
        spy = self

        def to(tensor, *args, **kwargs):
            out = spy._to(tensor, *args, **kwargs)
            if torch.is_tensor(out) and out.device != tensor.device:
                spy.events.append("to {} -> {}".format(tensor.device, out.device))
            return out

        def item(tensor):
            if tensor.is_cuda:
                spy.events.append("item of a device tensor")
            return spy._item(tensor)

        def tensor(data, *args, **kwargs):
            out = spy._tensor(data, *args, **kwargs)
            if out.is_cuda:
                spy.events.append("torch.tensor(..., device=cuda)")
            return out

        def is_all_true(valid):
            if valid.is_cuda:
                spy.events.append("_is_all_true of a device tensor")
            return spy._all(valid)

        torch.Tensor.to, torch.Tensor.item, torch.tensor, torch._is_all_true = to, item, tensor, is_all_true
        return self

    def __exit__(self, *exc):
        torch.Tensor.to, torch.Tensor.item, torch.tensor, torch._is_all_true = self._to, self._item, self._tensor, self._all
        return False


@pytest.mark.parametrize("family", ["lgssm1d", "gaussian"])
def test_reference_style_callables_do_not_talk_to_the_host(hip_device, family):
    """Inside `infer` a `Normal(mult * x, 0.5)` with default validate_args makes no host-to-device copy (the number is a
    cached device constant) and no `.all()` read (validation is deferred to the status word): the only host read of an
    evaluation is the status word at its end.  The numbers equal the same model evaluated with PyTorch's own
    `broadcast_all` / eager validation."""
    if family == "lgssm1d":
        parts, T, algorithm = lgssm1d(hip_device), 6, "smc"
    else:
        parts, T, algorithm = gaussian(hip_device), 1, "is"
    obs = observations_for(family, T, 16, hip_device)
    flags = dict(return_log_marginal_likelihood=True, return_latents=False)
    seed(5)
    with torch.no_grad():
        inference.infer(algorithm, obs, *parts, 32, **flags)       # fills the constant cache, loads the library
    seed(5)
    with torch.no_grad(), _CopySpy() as spy:
        fast = inference.infer(algorithm, obs, *parts, 32, **flags)["log_marginal_likelihood"]
    # the one read of the device status word at the end of `infer` is the only traffic with the host
    assert [e for e in spy.events if e != "item of a device tensor"] == [], spy.events
    assert spy.events.count("item of a device tensor") <= 1, spy.events
    seed(5)
    state.set_validation_mode("eager")
    try:
        with torch.no_grad():
            slow = inference.infer(algorithm, obs, *parts, 32, **flags)["log_marginal_likelihood"]
    finally:
        state.set_validation_mode("deferred")
    torch.testing.assert_close(fast, slow, rtol=0, atol=0)


def test_number_constants_are_cached_and_rounded_as_torch_tensor(hip_device):
    a = _syncfree.constant(0.7, torch.float32, hip_device)
    assert a is _syncfree.constant(0.7, torch.float32, hip_device)
    assert a.dim() == 0 and a.dtype == torch.float32 and a.device == hip_device
    assert float(a) == float(torch.tensor(0.7, dtype=torch.float32))
    assert _syncfree.constant(0.7, torch.float64, hip_device) is not a
    assert float(_syncfree.constant(1, torch.float32, hip_device)) == 1.0
    minus, plus = _syncfree.constant(-0.0, torch.float32, hip_device), _syncfree.constant(0.0, torch.float32, hip_device)
    assert np.signbit(float(minus)) and not np.signbit(float(plus))
    # outside `infer` torch.distributions is stock: the scale is a fresh tensor per call
    loc = torch.zeros(3, device=hip_device)
    outside = torch.distributions.Normal(loc, 0.7)
    assert outside.scale._base is None or outside.scale._base is not a
    with _syncfree.scope():
        inside = torch.distributions.Normal(loc, 0.7)
    assert inside.scale._base is a and inside.scale.shape == loc.shape
    torch.testing.assert_close(inside.scale, outside.scale, rtol=0, atol=0)


def test_invalid_parameters_surface_at_the_end_of_infer(hip_device):
    """Default validate_args=True is still honoured: a non-positive scale built inside a callable raises the reference's
    ValueError — at the end of `infer` (deferred), or at once with `set_validation_mode('eager')`."""
    initial, transition, emission, proposal = lgssm1d(hip_device)
    obs = observations_for("lgssm1d", 3, 4, hip_device)
    bad_scale = torch.tensor(-1.0, device=hip_device)

    def broken_emission(latents=None, time=None, previous_observations=None):
        return state.set_batch_shape_mode(torch.distributions.Normal(latents[-1], bad_scale),
                                          state.BatchShapeMode.FULLY_EXPANDED)

    with pytest.raises(ValueError, match="scale of Normal"):
        inference.infer("smc", obs, initial, transition, broken_emission, proposal, 8)
    inference.check_device_status(hip_device)      # the word was cleared when it was raised
    state.set_validation_mode("eager")
    try:
        with pytest.raises(ValueError, match="Expected parameter scale"):
            inference.infer("smc", obs, initial, transition, broken_emission, proposal, 8)
    finally:
        state.set_validation_mode("deferred")
    # validate_args=False: nobody checks (as in PyTorch); NaN log-weights are what surfaces
    def unchecked_emission(latents=None, time=None, previous_observations=None):
        return state.set_batch_shape_mode(torch.distributions.Normal(latents[-1], bad_scale, validate_args=False),
                                          state.BatchShapeMode.FULLY_EXPANDED)
    with pytest.raises(FloatingPointError):
        inference.infer("smc", obs, initial, transition, unchecked_emission, proposal, 8)


@pytest.mark.parametrize("family,algorithm", [("lgssm1d", "aesmc"), ("gaussian", "iwae")])
def test_reference_style_models_capture_and_replay_as_eager(hip_device, family, algorithm):
    """`GraphedLoss(backward=True)` of the literal reference classes: the capture succeeds (no host traffic inside
    the callables) and — its own verification aside — a replay from a seeded state equals the eager loss and gradients."""
    parts = lgssm1d(hip_device) if family == "lgssm1d" else gaussian(hip_device)
    T = 8 if family == "lgssm1d" else 1
    obs = observations_for(family, T, 32, hip_device)
    K = 64
    params = list(train.get_chained_params(*parts))
    seed(7)
    eager = losses.get_loss(obs, K, algorithm, *parts)
    eager.backward()
    want_loss, want_grads = eager.detach().clone(), [p.grad.clone() for p in params]
    del eager
    for p in params:
        p.grad = None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        warnings.filterwarnings("ignore", message="Inferred batch_shape_mode")
        graphed = graphs.GraphedLoss(obs, K, algorithm, *parts, backward=True, verify_replays=4,
                                     preserve_random_state=True)
    seed(7)
    loss = graphed()
    torch.testing.assert_close(loss, want_loss, rtol=1e-6, atol=1e-6)
    for p, want in zip(params, want_grads):
        torch.testing.assert_close(p.grad, want, rtol=1e-4, atol=1e-6)
    # fresh observations, fresh draws
    other = observations_for(family, T, 32, hip_device, seed=11)
    seed(8)
    with torch.no_grad():
        want = losses.get_loss(other, K, algorithm, *parts)
    seed(8)
    torch.testing.assert_close(graphed(other), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("family,algorithm", [("lgssm1d", "aesmc"), ("gaussian", "iwae")])
def test_train_captures_reference_style_models_without_a_warning(hip_device, family, algorithm):
    """`train.train` on the literal reference classes: the automatic capture (after 8 eager minibatches) succeeds —
    no "could not be captured" RuntimeWarning — and the run follows the eager loop's trajectory."""
    def run(hip_graph):
        seed(21)
        parts = lgssm1d(hip_device, 0.3, 0.5) if family == "lgssm1d" else gaussian(hip_device)
        true = lgssm1d(torch.device("cpu")) if family == "lgssm1d" else gaussian(torch.device("cpu"))
        T = 5 if family == "lgssm1d" else 1
        loader = train.get_synthetic_dataloader(true[0], true[1], true[2], T, 32)

        class OnDevice:
            def __iter__(self):
                return ([o.to(hip_device) for o in batch] for batch in loader)

        history = []
        with warnings.catch_warnings():
            # (the Gaussian family's untagged distributions warn about their inferred batch_shape_mode, as they do under
            #  the reference: state.py:24-58 — not what this test is about)
            warnings.filterwarnings("ignore", message="Inferred batch_shape_mode")
            warnings.filterwarnings("error", message=".*hipGraph.*")
            warnings.filterwarnings("error", message=".*host-side state.*")
            train.train(OnDevice(), 32, algorithm, *parts, num_epochs=1, num_iterations_per_epoch=24,
                        optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.01}, hip_graph=hip_graph,
                        reverify_every=6,
                        callback=lambda e, i, loss, *rest: history.append(float(loss.detach())))
        return history, [p.detach().clone() for p in train.get_chained_params(*parts)]

    eager_history, eager_params = run(False)
    auto_history, auto_params = run(None)
    assert len(auto_history) == 24
    np.testing.assert_allclose(auto_history, eager_history, rtol=1e-4)
    for got, want in zip(auto_params, eager_params):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-6)


def test_reverification_catches_host_state_frozen_into_the_graph(hip_device):
    """a Python-number coefficient changed from the callback is invisible to a replay.  Every
    `reverify_every`-th replay is compared with an eager evaluation: the change is noticed, ONE RuntimeWarning is
    raised and the loop is the eager loop from that step on."""
    initial, transition, emission, proposal = lgssm1d(hip_device)
    true = lgssm1d(torch.device("cpu"))
    loader = train.get_synthetic_dataloader(true[0], true[1], true[2], 4, 16)

    class OnDevice:
        def __iter__(self):
            return ([o.to(hip_device) for o in batch] for batch in loader)

    state_of_host = {"temperature": 1.0, "calls": 0}

    def annealed_emission(latents=None, time=None, previous_observations=None):
        state_of_host["calls"] += 1
        return state.set_batch_shape_mode(
            torch.distributions.Normal(emission.mult * latents[-1], 0.4 * state_of_host["temperature"]),
            state.BatchShapeMode.FULLY_EXPANDED)
    annealed_emission.__self__ = emission      # (its parameters are the emission module's)

    def callback(epoch, iteration, loss, *parts):
        if iteration == 5:
            state_of_host["temperature"] = 3.0

    seed(2)
    with pytest.warns(RuntimeWarning, match="host-side state has changed"):
        train.train(OnDevice(), 16, "aesmc", initial, transition, annealed_emission, proposal, num_epochs=1,
                    num_iterations_per_epoch=16, optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.01},
                    hip_graph=True, reverify_every=4, callback=callback)
    calls_after = state_of_host["calls"]
    assert calls_after > 0
    # ... and the callables are being CALLED again afterwards (the eager loop): one more minibatch, more calls
    seed(2)
    train.train(OnDevice(), 16, "aesmc", initial, transition, annealed_emission, proposal, num_epochs=1,
                num_iterations_per_epoch=1, optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.01},
                hip_graph=False)
    assert state_of_host["calls"] > calls_after


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_reference_training_run_replayed_through_the_captured_loop(hip_device, name):
    """tests/golden/train_*.npz — `aesmc.train.train` run by the reference on its own model classes, every draw
    recorded — reproduced with loss + backward of every minibatch issued as ONE hipGraph replay: the tape's noise goes
    into the capture's static noise buffers (`replay.StaticReplay`), its uniforms through the graph's uniform feed.
    Same bounds as the eager replay of the same fixtures (tests/test_gpu_configs_parity.py)."""
    case = Golden(name)
    meta = case.meta
    true, parts = _train_case_parts(meta, hip_device)
    named = {"{}.{}".format(part, pname): p for part, module in parts.items()
             if isinstance(module, torch.nn.Module) for pname, p in module.named_parameters()}
    with torch.no_grad():
        for pname, p in named.items():
            p.copy_(torch.from_numpy(case["init_" + pname]).to(hip_device))
    loader = train.get_synthetic_dataloader(*true, meta["num_timesteps"], meta["batch_size"])

    class OnDevice:
        def __iter__(self):
            return ([o.to(hip_device) for o in batch] for batch in loader)

    four = (parts["initial"], parts["transition"], parts["emission"], parts["proposal"])
    optimizer = torch.optim.SGD(train.get_chained_params(*four), lr=0.05)
    seen, graphed = [], None
    with warnings.catch_warnings(), replay.StaticReplay(case.tape()) as feed:
        warnings.simplefilter("ignore")
        feed.armed = True
        for epoch, iteration, observations in train._minibatches(OnDevice(), 2, 2):
            if graphed is None:
                feed.armed = False      # the capture's warm-up evaluations draw from the real generators
                graphed = graphs.GraphedLoss(observations, meta["num_particles"], meta["algorithm"], *four,
                                             backward=True, verify_replays=0)
                feed.armed = True
                assert len(feed.slots) > 0
            feed.load()
            loss = graphed(observations)
            optimizer.step()
            seen.append(float(loss))
    np.testing.assert_allclose(seen, case["losses"], rtol=1e-4)
    for pname, p in named.items():
        np.testing.assert_allclose(p.detach().cpu().numpy(), case["final_" + pname], rtol=1e-3, atol=1e-5)


def test_small_linear_layers_over_the_particles_run_as_k8_inside_infer(hip_device):
    """`nn.Linear(2, 1)` over B K rows — the reference's proposal net, test/models/lgssm.py:61-72 — is kernel K8 inside a
    sync-free scope (its backward K11), the library GEMM outside one and with `particle_linear` off: the same values and
    gradients to float32 rounding of another summation order."""
    from aesmc_amd import _kernels, settings
    provider = _kernels.get()
    torch.manual_seed(0)
    layer = torch.nn.Linear(2, 1).to(hip_device)
    rows = torch.randn(1 << 16, 2, device=hip_device, requires_grad=True)
    upstream = torch.randn(1 << 16, 1, device=hip_device)
    calls = {"k8": 0}
    real = provider.particle_affine

    def counting(*args, **kwargs):
        calls["k8"] += 1
        return real(*args, **kwargs)
    provider.particle_affine = counting
    try:
        def evaluate():
            for p in list(layer.parameters()) + [rows]:
                p.grad = None
            out = layer(rows)
            (out * upstream).sum().backward()
            return out.detach().clone(), rows.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()
        plain = evaluate()
        assert calls["k8"] == 0                      # outside `infer`: stock PyTorch
        with _syncfree.scope():
            fused = evaluate()
            assert calls["k8"] == 1
            with settings.override(particle_linear=False):
                off = evaluate()
            assert calls["k8"] == 1
            small = layer(rows[:64])                 # a handful of rows: not worth a launch of its own kind
            assert calls["k8"] == 1 and small.shape == (64, 1)
            three = layer(rows.view(4, -1, 2))       # [B, K, din] as the callables see particles
            assert calls["k8"] == 2 and three.shape == (4, (1 << 16) // 4, 1)
            torch.testing.assert_close(three.reshape(-1, 1), fused[0], rtol=0, atol=0)
    finally:
        provider.particle_affine = real
    for got, want, other in zip(fused, plain, off):
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
        torch.testing.assert_close(other, want, rtol=0, atol=0)
