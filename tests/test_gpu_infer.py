"""GPU parity tests, path level: aesmc_amd.inference.infer / losses.get_loss / train.train on the
MI355X against the fixtures captured from the reference (tests/golden/), with the reference's
random draws replayed on the device.

Stated tolerances (north star: indices bit-exact under a fixed seed, ELBO within fp32 tolerance):
  ancestor indices : exact.  For the one float32 fixture whose closest CDF comparison sits 8e-7
                     from flipping (meta["margin"]) end-to-end float32 rounding on a different
                     device may legitimately move it, so that case demands >= 99.9 % agreement end
                     to end AND exact equality when the kernel is fed the reference's own
                     log-weights (teacher forcing);
  log-weights      : float32 rtol 1e-5 / atol 1e-5, float64 1e-11;
  log Z, loss      : float32 |d| <= 1e-4 (1 + |x|), float64 1e-10;
  gradients        : float32 1e-3 of the largest entry, float64 1e-8.
"""
import numpy as np
import pytest
import torch

from aesmc_amd import inference, losses, state, statistics, train
from aesmc_amd import math as amath
from aesmc_amd.testing import models, replay
from tests.golden_io import Golden, INFER_CASES

pytestmark = pytest.mark.gpu

Normal = torch.distributions.Normal
Modes = state.BatchShapeMode


def test_native_library_is_loaded_and_is_the_backend(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    state.resample(torch.zeros(1, 2, device=hip_device), torch.zeros(1, 2, dtype=torch.int64, device=hip_device))
    with open("/proc/self/maps") as fh:
        assert "libaesmc_hip.so" in fh.read()


def tolerances(dtype):
    if dtype == torch.float32:
        return dict(lw=dict(rtol=1e-5, atol=1e-5), lml=1e-4, grad=1e-3)
    return dict(lw=dict(rtol=1e-11, atol=1e-11), lml=1e-10, grad=1e-8)


def run(case, device, **flags):
    parts, named = case.build_parts(state, device)
    observations = case.observations(device)
    smc = case.meta["algorithm"] == "aesmc"
    with replay.replay(case.tape()):
        result = inference.infer("smc" if smc else "is", observations, parts["initial"],
                                 parts["transition"], parts["emission"], parts["proposal"],
                                 case.meta["num_particles"], **flags)
    return result, parts, named, observations


@pytest.mark.parametrize("name", INFER_CASES)
def test_infer_matches_reference_fixture(hip_device, name):
    case = Golden(name)
    tol = tolerances(case.dtype)
    smc = case.meta["algorithm"] == "aesmc"
    result, parts, named, observations = run(
        case, hip_device, return_log_marginal_likelihood=True, return_latents=True,
        return_original_latents=smc, return_log_weights=True, return_ancestral_indices=smc)
    fragile = smc and case.dtype == torch.float32 and case.meta.get("margin", 1.0) < 5e-6
    if smc:
        got_idx = [a.cpu().numpy() for a in result["ancestral_indices"]]
        assert all(a.dtype == torch.int64 and a.is_cuda for a in result["ancestral_indices"])
        want_idx = case.series("out_idx")
        agreement = np.mean([(g == w).mean() for g, w in zip(got_idx, want_idx)])
        if fragile:
            assert agreement >= 0.999, agreement
        else:
            for g, w in zip(got_idx, want_idx):
                np.testing.assert_array_equal(g, w)
        # teacher forcing: reference log-weights + reference uniforms -> reference indices, exactly
        from aesmc_amd import _ops
        for t, want in enumerate(want_idx):
            lw = torch.from_numpy(case["out_log_weights_{}".format(t)]).to(hip_device)
            u = torch.from_numpy(case["uniform_{}".format(t)].reshape(-1)).to(hip_device)
            np.testing.assert_array_equal(_ops.ancestor_index(lw, u).cpu().numpy(), want)
        exact = all((g == w).all() for g, w in zip(got_idx, want_idx))
    else:
        exact = True
    if exact:  # past the first flipped ancestor the two runs are different (valid) particle systems
        for got, want in zip(result["log_weights"], case.series("out_log_weights")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol["lw"])
        for got, want in zip(result["latents"], case.series("out_latents")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol["lw"])
        np.testing.assert_allclose(result["last_latent"].detach().cpu().numpy(), case["out_last_latent"], **tol["lw"])
        if smc:
            for got, want in zip(result["original_latents"], case.series("out_original_latents")):
                np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol["lw"])
    lml = result["log_marginal_likelihood"].detach().cpu().numpy()
    want = case["out_lml"]
    bound = (tol["lml"] if exact else 1e-2) * (1 + np.abs(want))
    assert (np.abs(lml - want) <= bound).all(), (lml, want)

    with replay.replay(case.tape()):
        loss = losses.get_loss(observations, case.meta["num_particles"], case.meta["algorithm"],
                               parts["initial"], parts["transition"], parts["emission"], parts["proposal"])
    loss.backward()
    want_loss = float(case["out_loss"])
    assert abs(loss.item() - want_loss) <= (tol["lml"] if exact else 1e-2) * (1 + abs(want_loss))
    if exact:
        for pname, p in named.items():
            want = case["grad_" + pname]
            scale = np.abs(want).max() + 1e-30
            np.testing.assert_allclose(p.grad.cpu().numpy() / scale, want / scale, rtol=0, atol=tol["grad"])


def test_smc_estimate_is_unbiased_against_kalman_filter(hip_device):
    """Statistical pin independent of any fixture: E[Z_hat] == exact LGSSM likelihood."""
    torch.manual_seed(0)
    np.random.seed(0)
    model = models.LgssmNd(2, seed=0, dtype=torch.float64, validate_args=False).to(hip_device)
    observations = model.simulate(5, 1, seed=2)
    exact = models.kalman_log_likelihood(model, observations)[0]
    repeated = [o.expand(64, -1).contiguous() for o in observations]   # 64 independent runs as a batch
    with torch.no_grad():
        out = inference.infer("smc", repeated, model.initial, model.transition, model.emission,
                              model.proposal, 2000, return_log_marginal_likelihood=True,
                              return_latents=False)
    estimates = out["log_marginal_likelihood"].cpu().numpy()
    log_mean = np.log(np.mean(np.exp(estimates - exact))) + exact
    assert abs(log_mean - exact) < 0.05, (log_mean, exact)


def test_full_size_config2_runs_and_matches_cpu_port_statistically(hip_device):
    """configs[1] of BASELINE.json at full size: finite ELBO, sorted in-range ancestors, the
    batch-mean ELBO within Monte-Carlo error of the Kalman likelihood bound."""
    B, K, T, d = 256, 1024, 50, 10
    model = models.LgssmNd(d, seed=0, validate_args=False).to(hip_device)
    observations = model.simulate(T, B, seed=1)
    np.random.seed(0)
    torch.manual_seed(0)
    with torch.no_grad():
        out = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                              model.proposal, K, return_log_marginal_likelihood=True,
                              return_latents=False, return_ancestral_indices=True)
    lml = out["log_marginal_likelihood"]
    assert lml.shape == (B,) and bool(torch.isfinite(lml).all())
    for idx in out["ancestral_indices"]:
        assert idx.shape == (B, K)
        assert bool((idx[:, 1:] >= idx[:, :-1]).all()) and int(idx.min()) >= 0 and int(idx.max()) < K
    # log Z_hat is a downward-biased (Jensen), noisy estimate of the exact log-likelihood: with the
    # untrained proposal of this workload the gap is a few nats over T = 50 steps in d = 10.
    exact = models.kalman_log_likelihood(model, [o[:16] for o in observations])
    gap = lml[:16].cpu().numpy() - exact
    assert gap.max() < 3.0 and -15.0 < gap.mean() < 0.5, gap


def test_deferred_errors_surface_at_the_end_of_infer(hip_device):
    model = models.LgssmNd(2, seed=0, validate_args=False).to(hip_device)
    observations = model.simulate(3, 2, seed=0)

    def nan_emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        return state.set_batch_shape_mode(Normal(dist.loc * float("nan"), 1.0, validate_args=False),
                                          Modes.FULLY_EXPANDED)

    with pytest.raises(FloatingPointError):      # aesmc/inference.py:244-245
        inference.infer("smc", observations, model.initial, model.transition, nan_emission, model.proposal, 8)
    with pytest.raises(FloatingPointError):
        inference.sample_ancestral_index(torch.tensor([[0.0, float("nan")]], device=hip_device))

    def dead_emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        return state.set_batch_shape_mode(Normal(dist.loc, 1e-30, validate_args=False), Modes.FULLY_EXPANDED)

    with pytest.raises(RuntimeError):            # index K out of range, as torch.gather would raise
        inference.infer("smc", [1e6 * o for o in observations], model.initial, model.transition,
                        dead_emission, model.proposal, 8)
    # the status word is clean again: a healthy call right after succeeds
    out = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                          model.proposal, 8, return_log_marginal_likelihood=True)
    assert bool(torch.isfinite(out["log_marginal_likelihood"]).all())


def test_validation_modes(hip_device):
    gamma = torch.distributions.Gamma(torch.ones(2, 3, device=hip_device), 1.0, validate_args=False)
    bad = -torch.ones(2, 3, device=hip_device)
    from aesmc_amd import _kernels, _lib
    _kernels.get().read_flags(hip_device)
    state.log_prob(gamma, bad)                   # deferred: no raise here ...
    assert _kernels.get().read_flags(hip_device) == _lib.FLAG_VALUE_OUTSIDE_SUPPORT   # ... flagged
    with pytest.raises(ValueError):              # shapes are still checked at once
        state.log_prob(Normal(torch.zeros(2, 3, device=hip_device), 1.0), torch.zeros(2, 4, device=hip_device))
    state.set_validation_mode("eager")
    try:
        with pytest.raises(ValueError):
            state.log_prob(gamma, bad)
    finally:
        state.set_validation_mode("deferred")


def test_sample_ancestral_index_on_device(hip_device):
    out = inference.sample_ancestral_index(torch.rand(4, 9, device=hip_device))
    assert out.is_cuda and out.dtype == torch.int64 and out.shape == (4, 9)
    np.random.seed(1)
    lw = torch.randn(16, 300, device=hip_device, dtype=torch.float64)
    with replay.record() as tape:
        got = inference.sample_ancestral_index(lw)
    from oracle import kernel_oracle
    want, _ = kernel_oracle.ancestor_index(lw.cpu().numpy(), tape.uniforms[0].reshape(-1))
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_math_and_statistics_on_device(hip_device):
    x = torch.randn(3, 5, 7, device=hip_device, dtype=torch.float64)
    for dim in (0, 1, 2):
        torch.testing.assert_close(amath.lognormexp(x, dim=dim), x - torch.logsumexp(x, dim, keepdim=True))
        torch.testing.assert_close(amath.exponentiate_and_normalize(x, dim=dim), torch.softmax(x, dim))
    arr = np.array([1.0, 2.0, 3.0])              # test/test_math.py:51-64, numpy branch
    want = np.log(np.exp(arr) / np.exp(arr).sum())
    got = amath.lognormexp(arr)
    assert isinstance(got, np.ndarray)
    np.testing.assert_allclose(got, want, atol=1e-6)
    np.testing.assert_allclose(amath.lognormexp(np.array([1, 2, 3])), want, atol=1e-6)
    np.testing.assert_allclose(amath.exponentiate_and_normalize(arr), np.exp(want), rtol=1e-7)
    lw = torch.randn(4, 50, device=hip_device, dtype=torch.float64)
    w = torch.softmax(lw, 1)
    torch.testing.assert_close(statistics.ess(lw), 1.0 / (w ** 2).sum(1))
    value = torch.randn(4, 50, 3, device=hip_device, dtype=torch.float64)
    torch.testing.assert_close(statistics.empirical_mean(value, lw), (w[..., None] * value).sum(1))
    xg = x.clone().requires_grad_()
    amath.lognormexp(xg, dim=1).exp().sum().backward()      # autograd through K1's backward
    assert float(xg.grad.abs().max()) < 1e-9                # d/dx sum softmax == 0


def test_autograd_through_the_hot_path_matches_torch(hip_device):
    """K3 and K1 backward against autograd of the torch primitives they replace."""
    from aesmc_amd import _ops
    B, K, d = 5, 200, 4
    gen = torch.Generator(device=hip_device).manual_seed(0)
    value = torch.randn(B, K, d, device=hip_device, dtype=torch.float64, generator=gen)
    terms = [torch.randn(B, K, device=hip_device, dtype=torch.float64, generator=gen) for _ in range(3)]
    idx = _ops.ancestor_index(terms[0], torch.rand(B, device=hip_device, dtype=torch.float64, generator=gen))

    def objective(gather, combine):
        v = value.clone().requires_grad_()
        ts = [t.clone().requires_grad_() for t in terms]
        lw, lse = combine(*ts)
        loss = (gather(v, idx).sum(-1) * lw).sum() + (lse ** 2).sum()
        loss.backward()
        return loss.detach(), [v.grad] + [t.grad for t in ts]

    mine = objective(_ops.resample_gather, _ops.logweight_lse)
    ref = objective(lambda v, i: torch.gather(v, 1, i[..., None].expand_as(v)),
                    lambda a, b, c: (a + b - c, torch.logsumexp(a + b - c, 1)))
    torch.testing.assert_close(mine[0], ref[0], rtol=1e-12, atol=1e-12)
    for g, w in zip(mine[1], ref[1]):
        torch.testing.assert_close(g, w, rtol=1e-10, atol=1e-10)


def test_train_on_device(hip_device):
    """test/test_losses.py:11-79 shape of test: the whole train -> loss -> infer -> backward chain."""
    torch.manual_seed(0)
    np.random.seed(0)
    prior = models.GaussianPrior(0.0, 1.0).to(hip_device)
    likelihood = models.GaussianLikelihood(1.0).to(hip_device)
    network = models.GaussianInferenceNetwork(0.1, 0.0, 1.5).to(hip_device)
    true_prior = models.GaussianPrior(1.0, 1.0).to(hip_device)
    true_likelihood = models.GaussianLikelihood(0.5).to(hip_device)
    loader = train.get_synthetic_dataloader(true_prior, None, true_likelihood, 1, 64)
    history = []
    train.train(loader, 16, "iwae", prior, None, likelihood, network, num_epochs=1,
                num_iterations_per_epoch=80, optimizer_algorithm=torch.optim.SGD,
                optimizer_kwargs={"lr": 0.05},
                callback=lambda e, i, loss, *parts: history.append(loss.item()))
    assert len(history) == 80 and np.isfinite(history).all()
    assert np.mean(history[-10:]) < np.mean(history[:10])
    # SMC training step on the d-dim LGSSM: gradients reach every parameter
    model = models.LgssmNd(3, seed=0, validate_args=False).to(hip_device)
    loss = losses.get_loss(model.simulate(6, 8, seed=1), 64, "aesmc", model.initial, model.transition,
                           model.emission, model.proposal)
    loss.backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


def test_dict_latents_and_observations_end_to_end(hip_device):
    """The reference's helpers recurse through dicts (state.py:93, 168-170, 199-200) but its
    log_prob dict branch is dead code (state.py:130); here a model whose latent and observation are
    dicts {'a', 'b'} must give exactly the numbers of the same model written with one tensor."""
    B, K, T = 3, 32, 4
    torch.manual_seed(0)
    y = [torch.randn(B, 2, device=hip_device, dtype=torch.float64) for _ in range(T)]
    scale = torch.tensor(0.9, device=hip_device, dtype=torch.float64)
    full = Modes.FULLY_EXPANDED

    def tag(dist, mode):
        return state.set_batch_shape_mode(dist, mode)

    def tensor_model():
        def initial():
            return tag(Normal(torch.zeros(2, device=hip_device, dtype=torch.float64), scale), Modes.NOT_EXPANDED)

        def transition(previous_latents=None, time=None, previous_observations=None):
            return tag(Normal(0.8 * previous_latents[-1], scale), full)

        def emission(latents=None, time=None, previous_observations=None):
            return tag(Normal(latents[-1], scale), full)

        def proposal(previous_latents=None, time=None, observations=None):
            if time == 0:
                return tag(Normal(0.5 * observations[0], scale), Modes.BATCH_EXPANDED)
            return tag(Normal(0.4 * previous_latents[-1] + 0.5 * observations[time].unsqueeze(1), scale), full)

        return initial, transition, emission, proposal, y

    def dict_model():
        def split(t):
            return {"a": t[..., 0], "b": t[..., 1]}

        def initial():
            zero = torch.zeros((), device=hip_device, dtype=torch.float64)
            return {k: tag(Normal(zero, scale), Modes.NOT_EXPANDED) for k in ("a", "b")}

        def transition(previous_latents=None, time=None, previous_observations=None):
            return {k: tag(Normal(0.8 * previous_latents[-1][k], scale), full) for k in ("a", "b")}

        def emission(latents=None, time=None, previous_observations=None):
            return {k: tag(Normal(latents[-1][k], scale), full) for k in ("a", "b")}

        def proposal(previous_latents=None, time=None, observations=None):
            if time == 0:
                return {k: tag(Normal(0.5 * observations[0][k], scale), Modes.BATCH_EXPANDED) for k in ("a", "b")}
            return {k: tag(Normal(0.4 * previous_latents[-1][k] + 0.5 * observations[time][k].unsqueeze(1), scale), full)
                    for k in ("a", "b")}

        return initial, transition, emission, proposal, [split(o) for o in y]

    def run(make, noise):
        initial, transition, emission, proposal, observations = make()
        with replay.replay(noise):
            return inference.infer("smc", observations, initial, transition, emission, proposal, K,
                                   return_log_marginal_likelihood=True, return_log_weights=True,
                                   return_ancestral_indices=True, return_latents=True)

    gen = torch.Generator().manual_seed(1)
    eps = [torch.randn(K, B, 2, generator=gen, dtype=torch.float64).numpy()] + \
        [torch.randn(B, K, 2, generator=gen, dtype=torch.float64).numpy() for _ in range(T - 1)]
    uniforms = [np.random.RandomState(t).uniform(size=(B, 1)) for t in range(T - 1)]
    per_key = []
    for block in eps:                       # dict proposals draw key 'a' then key 'b'
        per_key += [np.ascontiguousarray(block[..., 0]), np.ascontiguousarray(block[..., 1])]
    as_tensor = run(tensor_model, replay.Tape(eps, uniforms))
    as_dict = run(dict_model, replay.Tape(per_key, uniforms))
    for a, b in zip(as_tensor["ancestral_indices"], as_dict["ancestral_indices"]):
        assert torch.equal(a, b)
    for a, b in zip(as_tensor["log_weights"], as_dict["log_weights"]):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(as_tensor["log_marginal_likelihood"], as_dict["log_marginal_likelihood"],
                               rtol=1e-12, atol=1e-12)
    for a, b in zip(as_tensor["latents"], as_dict["latents"]):
        torch.testing.assert_close(a[..., 0], b["a"])
        torch.testing.assert_close(a[..., 1], b["b"])


def test_fused_normal_log_prob_equals_eager_path(hip_device):
    """state.log_prob through kernel K4 vs the reference's eager Normal.log_prob route, forward
    and backward, for every BatchShapeMode."""
    B, K, d = 3, 40, 5
    gen = torch.Generator(device=hip_device).manual_seed(0)
    value = torch.randn(B, K, d, device=hip_device, dtype=torch.float64, generator=gen)
    weights = torch.randn(B, K, device=hip_device, dtype=torch.float64, generator=gen)
    params = {"full": torch.randn(B, K, d, device=hip_device, dtype=torch.float64, generator=gen),
              "batch": torch.randn(B, d, device=hip_device, dtype=torch.float64, generator=gen),
              "none": torch.randn(d, device=hip_device, dtype=torch.float64, generator=gen)}
    log_scale = torch.zeros(d, device=hip_device, dtype=torch.float64)
    for kind, loc0 in params.items():
        results = []
        for fused in (True, False):
            state.set_fused_normal(fused)
            try:
                v = value.clone().requires_grad_()
                loc = loc0.clone().requires_grad_()
                ls = log_scale.clone().requires_grad_()
                dist = Normal(loc, ls.exp() * 0.8)
                out = state.log_prob(dist, v)
                (out * weights).sum().backward()
                results.append((out.detach(), v.grad, loc.grad, ls.grad))
            finally:
                state.set_fused_normal(True)
        for a, b in zip(*results):
            torch.testing.assert_close(a, b, rtol=1e-11, atol=1e-11)
    # Independent(Normal) and Python-number parameters take the fused route too
    ind = torch.distributions.Independent(Normal(params["full"], 0.5), 1)
    torch.testing.assert_close(state.log_prob(ind, value), ind.log_prob(value))
    flat = value.float()[..., 0].contiguous()
    torch.testing.assert_close(state.log_prob(Normal(0.0, 1.0), flat), Normal(0.0, 1.0).log_prob(flat))
    with pytest.raises(RuntimeError):   # three missing batch dims: rejected like state.py:146-150
        state.log_prob(Normal(0.0, 1.0), value.float())


@pytest.mark.parametrize("B,K,T,d", [(3, 1, 4, 2), (1, 7, 3, 1), (2, 7, 3, 3), (2, 1023, 3, 5), (5, 64, 1, 3),
                                     (4, 256, 6, 4), (2, 33000, 3, 2), (1, 4096, 2, 16)])
@pytest.mark.parametrize("history", ["lazy", "eager"])
def test_odd_shapes_match_the_cpu_port_draw_for_draw(hip_device, B, K, T, d, history):
    """The op-for-op CPU port of the reference (oracle/reference_port.py) run on CPU with its random
    draws recorded; the product on the GPU replays them.  float64: indices exact, log-weights and
    log Z to 1e-10.  Shapes chosen to cross every dispatch boundary of the resampling launch: one
    particle, payload rows the fused step declines (K * row_bytes not a multiple of 16), T = 1,
    more particles than one workgroup holds (K2 over the workspace + K1 + K3), wide rows."""
    from oracle import reference_port
    dtype = torch.float64
    cpu_model = models.LgssmNd(d, seed=0, dtype=dtype, state=reference_port)
    observations = cpu_model.simulate(T, B, seed=1)
    np.random.seed(3)
    torch.manual_seed(3)
    flags = dict(return_log_marginal_likelihood=True, return_log_weights=True, return_ancestral_indices=True,
                 return_original_latents=True)
    with replay.record() as tape:
        want = reference_port.infer("smc", observations, cpu_model.initial, cpu_model.transition,
                                    cpu_model.emission, cpu_model.proposal, K, **flags)
    model = models.LgssmNd(d, seed=0, dtype=dtype).to(hip_device)
    inference.set_history_mode(history)
    try:
        with replay.replay(tape):
            got = inference.infer("smc", [o.to(hip_device) for o in observations], model.initial,
                                  model.transition, model.emission, model.proposal, K, **flags)
    finally:
        inference.set_history_mode("lazy")
    assert len(got["ancestral_indices"]) == T - 1
    for a, b in zip(got["ancestral_indices"], want["ancestral_indices"]):
        assert torch.equal(a.cpu(), b)
    for a, b in zip(got["log_weights"], want["log_weights"]):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-10, atol=1e-10)
    for a, b in zip(got["latents"], want["latents"]):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(got["log_marginal_likelihood"].cpu(), want["log_marginal_likelihood"],
                               rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(got["last_latent"].cpu(), want["last_latent"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kind,algorithm,B,K,T,d", [("nonlinear", "aesmc", 4, 256, 5, 6), ("nonlinear", "iwae", 3, 64, 3, 4),
                                                    ("learned_scale", "aesmc", 4, 256, 5, 6), ("learned_scale", "iwae", 3, 64, 3, 4),
                                                    ("iwae", "iwae", 64, 512, 1, 1), ("lgssm", "aesmc", 3, 300, 4, 5)])
def test_other_model_families_match_the_cpu_port_with_gradients(hip_device, kind, algorithm, B, K, T, d):
    """BASELINE.json's other configs (nonlinear SSM + MLP proposal; one-step Gaussian IWAE) and the
    LGSSM once more, through get_loss + backward: the CPU port records its draws, the GPU replays
    them; float64: loss to 1e-10, every parameter gradient to 1e-8 of its largest entry."""
    from oracle import reference_port
    dtype = torch.float64

    def build(state_module, device):
        if kind == "nonlinear":
            model = models.NonlinearSsm(d, hidden=16, seed=0, dtype=dtype, state=state_module)
        elif kind == "learned_scale":     # tensor scales from the proposal net, a learned vector scale
            model = models.LearnedScaleSsm(d, hidden=16, seed=0, dtype=dtype, state=state_module)
        elif kind == "iwae":
            model = models.GaussianIwae(dtype=dtype, state=state_module)
        else:
            model = models.LgssmNd(d, seed=0, dtype=dtype, state=state_module)
        return model.to(device)

    cpu_model = build(reference_port, torch.device("cpu"))
    observations = cpu_model.simulate(T, B, seed=1)
    parts = lambda m: (m.initial, m.transition, m.emission, m.proposal)
    np.random.seed(4)
    torch.manual_seed(4)
    with replay.record() as tape:
        want = reference_port.get_loss(observations, K, algorithm, *parts(cpu_model))
    want.backward()
    model = build(state, hip_device)
    with replay.replay(tape):
        got = losses.get_loss([o.to(hip_device) for o in observations], K, algorithm, *parts(model))
    got.backward()
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-10, atol=1e-10)
    for (name, p), q in zip(model.named_parameters(), cpu_model.parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        scale = float(q.grad.abs().max()) + 1e-300
        torch.testing.assert_close(p.grad.cpu() / scale, q.grad / scale, rtol=0, atol=1e-8, msg=name)


@pytest.mark.parametrize("seed_", [1, 2, 3])
def test_random_configurations_match_the_cpu_port(hip_device, seed_):
    """Fuzz: 20 random (model family, algorithm, history mode, B, K, T, d) combinations per seed —
    LGSSM, nonlinear SSM and the learned-scale SSM; K from 1 to 3000 across every chunk-size switch
    of the resampling kernel — loss and every parameter gradient against the CPU port with its
    recorded draws replayed (float64: loss 1e-10, gradients 1e-7 of the largest entry)."""
    from oracle import reference_port
    rng = np.random.RandomState(seed_)
    dtype = torch.float64
    for case in range(20):
        B, T = int(rng.randint(1, 7)), int(rng.randint(1, 7))
        K = int(rng.choice([1, 2, 5, 16, 63, 64, 65, 200, 513, 1024, 3000]))
        d = int(rng.choice([1, 2, 3, 4, 10, 16]))
        kind = rng.choice(["lgssm", "nonlinear", "learned_scale"])
        algorithm = rng.choice(["aesmc", "iwae"])
        history = rng.choice(["lazy", "eager"])

        def build(state_module, device):
            cls = {"lgssm": models.LgssmNd, "nonlinear": models.NonlinearSsm,
                   "learned_scale": models.LearnedScaleSsm}[kind]
            extra = {} if kind == "lgssm" else {"hidden": 8}
            return cls(d, seed=case, dtype=dtype, state=state_module, **extra).to(device)

        cpu_model = build(reference_port, torch.device("cpu"))
        observations = cpu_model.simulate(T, B, seed=case)
        parts = lambda m: (m.initial, m.transition, m.emission, m.proposal)
        np.random.seed(case)
        torch.manual_seed(case)
        with replay.record() as tape:
            want = reference_port.get_loss(observations, K, algorithm, *parts(cpu_model))
        want.backward()
        model = build(state, hip_device)
        inference.set_history_mode(history)
        try:
            with replay.replay(tape):
                got = losses.get_loss([o.to(hip_device) for o in observations], K, algorithm, *parts(model))
            got.backward()
        finally:
            inference.set_history_mode("lazy")
        label = (seed_, case, kind, algorithm, history, B, K, T, d)
        torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-10, atol=1e-10, msg=str(label))
        for (name, p), q in zip(model.named_parameters(), cpu_model.parameters()):
            if q.grad is None:
                continue
            scale = float(q.grad.abs().max()) + 1e-300
            torch.testing.assert_close(p.grad.cpu() / scale, q.grad / scale, rtol=0, atol=1e-7, msg=str(label + (name,)))


def test_status_word_hygiene(hip_device):
    """(1) the stand-alone resampler raises for EVERY flag, not only NaN — an all -inf row
    yields indices equal to num_particles, where the reference fails inside np.digitize;
    (2) stand-alone helpers only SET flags and `inference.check_device_status` reads and raises;
    (3) an `infer` abandoned by an exception out of a user callable takes what its kernels had
    flagged with it: the next, healthy call does not raise a stale error."""
    dead = torch.full((2, 8), float("-inf"), device=hip_device)
    with pytest.raises(RuntimeError):
        inference.sample_ancestral_index(dead)
    inference.check_device_status(hip_device)                       # clean again

    value = torch.arange(24, dtype=torch.float32, device=hip_device).reshape(2, 4, 3)
    state.resample(value, torch.tensor([[0, 1, 9, 3], [0, 0, 0, 2]], device=hip_device))   # clamps, flags, no raise
    with pytest.raises(RuntimeError):
        inference.check_device_status(hip_device)
    inference.check_device_status(hip_device)

    model = models.LgssmNd(2, seed=0, validate_args=False).to(hip_device)
    observations = model.simulate(4, 3, seed=0)

    class Boom(Exception):
        pass

    def bad_emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        if time == 0:      # NaN log-weights at time 0: flagged by the resampling launch of time 1
            return state.set_batch_shape_mode(Normal(dist.loc * float("nan"), 1.0, validate_args=False),
                                              Modes.FULLY_EXPANDED)
        if time == 2:
            torch.cuda.synchronize()
            raise Boom()
        return dist

    with pytest.raises(Boom):
        inference.infer("smc", observations, model.initial, model.transition, bad_emission, model.proposal, 16)
    out = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal, 16,
                          return_log_marginal_likelihood=True)
    assert bool(torch.isfinite(out["log_marginal_likelihood"]).all())


def test_genealogy_gradients_take_the_sorted_kernel_and_match_torch(hip_device):
    """`get_resampled_latents` (aesmc/inference.py:196-231) on K2's own indices: the composed lineage
    is tagged sorted, so the backward of its T gathers is the atomic-free segmented sum (bitwise
    reproducible from run to run), the promise holds (no UNSORTED flag) and the gradients equal
    autograd through torch.gather on the same composition; untagged indices (the reference's own
    genealogy test uses unsorted ones) keep the order-agnostic kernel and the same values."""
    from aesmc_amd import _kernels, _ops
    B, K, d, T = 3, 300, 5, 4
    gen = torch.Generator(device=hip_device).manual_seed(1)
    rand = lambda *shape: torch.randn(*shape, device=hip_device, dtype=torch.float64, generator=gen)
    latents = [rand(B, K, d) for _ in range(T)]
    weights = [rand(B, K, d) for _ in range(T)]
    indices = [_ops.ancestor_index(rand(B, K), torch.rand(B, device=hip_device, dtype=torch.float64, generator=gen))
               for _ in range(T - 1)]
    assert all(getattr(i, "_aesmc_sorted", False) for i in indices)

    def run(index_list):
        leaves = [x.clone().requires_grad_() for x in latents]
        out = inference.get_resampled_latents(leaves, index_list)
        sum((o * w).sum() for o, w in zip(out, weights)).backward()
        return [o.detach() for o in out], [x.grad for x in leaves]

    values, grads = run(indices)
    again_values, again_grads = run(indices)
    untagged = [i.clone() for i in indices]                     # clones carry no tag
    assert not any(getattr(i, "_aesmc_sorted", False) for i in untagged)
    other_values, other_grads = run(untagged)
    assert _kernels.get().read_flags(hip_device) == 0
    # torch reference: compose the lineage with torch.gather, gather every latent, autograd
    leaves = [x.clone().requires_grad_() for x in latents]
    lineage = torch.arange(K, device=hip_device).unsqueeze(0).expand(B, K)
    total = 0
    want_values = [None] * T
    for t in range(T - 1, -1, -1):
        want_values[t] = torch.gather(leaves[t], 1, lineage[..., None].expand(B, K, d))
        total = total + (want_values[t] * weights[t]).sum()
        if t > 0:
            lineage = torch.gather(indices[t - 1], 1, lineage)
    total.backward()
    for t in range(T):
        assert torch.equal(values[t], want_values[t].detach())
        assert torch.equal(again_grads[t], grads[t])            # no atomics: same bits twice
        torch.testing.assert_close(grads[t], leaves[t].grad, rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(other_grads[t], leaves[t].grad, rtol=1e-12, atol=1e-12)


def test_a_falsely_tagged_index_leaves_zero_rows_and_a_flag_not_stale_memory(hip_device):
    """the sorted-index backward writes each row of the gradient exactly once IF the indices are
    non-decreasing.  A tag that was inherited (here: set by hand on an unsorted index) starts from a zeroed
    gradient: the result is finite (zeros where no tile wrote), the flag is raised, and K2's own outputs — sorted by
    construction — keep the path without the fill."""
    from aesmc_amd import _kernels, _lib, _ops
    provider = _kernels.get()
    provider.read_flags(hip_device)
    B, K, d = 2, 700, 5
    gen = torch.Generator(device=hip_device).manual_seed(3)
    grad = torch.randn(B, K, d, device=hip_device, generator=gen)
    index = torch.randint(0, K, (B, K), device=hip_device, generator=gen)        # not sorted
    index._aesmc_sorted = "inherited"
    poison = torch.full((B, K, d), float("nan"), device=hip_device)               # what recycled memory may hold
    del poison
    first = provider.gather_backward(grad, index, sorted_index="inherited")
    assert provider.read_flags(hip_device) & _lib.FLAG_UNSORTED_INDEX
    second = provider.gather_backward(grad, index, sorted_index="inherited")
    assert provider.read_flags(hip_device) & _lib.FLAG_UNSORTED_INDEX
    # (finite, not reproducible: tiles of an unsorted index claim overlapping destination ranges — the flag says so)
    assert bool(torch.isfinite(first).all()) and bool(torch.isfinite(second).all())
    # through autograd the tag travels with the index tensor
    value = torch.randn(B, K, d, device=hip_device, generator=gen, requires_grad=True)
    _ops.resample_gather(value, index).sum().backward()
    assert bool(torch.isfinite(value.grad).all())
    assert provider.read_flags(hip_device) & _lib.FLAG_UNSORTED_INDEX
    # a correct inherited tag: the same numbers as the untagged kernel
    lw = torch.randn(B, K, device=hip_device, dtype=torch.float64, generator=gen)
    own = _ops.ancestor_index(lw, torch.rand(B, device=hip_device, dtype=torch.float64, generator=gen))
    inherited = own.clone()
    inherited._aesmc_sorted = "inherited"
    a = provider.gather_backward(grad, own, sorted_index=True)
    b = provider.gather_backward(grad, inherited, sorted_index="inherited")
    c = provider.gather_backward(grad, own.clone(), sorted_index=False)
    assert torch.equal(a, b)
    torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-5)
    assert provider.read_flags(hip_device) == 0
