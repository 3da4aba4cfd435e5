"""CPU tests of aesmc_amd's HOST logic (shape modes, the infer loop, error behaviour, lazy
history, training glue).  Kernels are substituted by the NumPy oracle through the `oracle_backend`
fixture — a test hook; the product itself refuses CPU tensors (see test_library.py).

Structure follows the reference's own suites (test/test_state.py, test/test_inference.py,
test/test_losses.py, test/test_statistics.py), cited per test.
"""
import warnings

import numpy as np
import pytest
import torch

import aesmc_amd
from aesmc_amd import inference, losses, state, statistics, train
from aesmc_amd import math as amath
from aesmc_amd.testing import models, replay
from tests.golden_io import Golden, INFER_CASES

Normal = torch.distributions.Normal
Modes = state.BatchShapeMode


# ---- state: batch shape modes (test/test_state.py:7-52) ------------------------------------------
def test_batch_shape_mode_explicit_and_inferred():
    B, K = 2, 3
    dist = state.set_batch_shape_mode(Normal(torch.zeros(B, K), 1.0), Modes.FULLY_EXPANDED)
    assert state.get_batch_shape_mode(dist) == Modes.FULLY_EXPANDED
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # unambiguous cases must not warn
        assert state.get_batch_shape_mode(Normal(0.0, 1.0), B, K) == Modes.NOT_EXPANDED
        assert state.get_batch_shape_mode(Normal(torch.zeros(5), 1.0), B, K) == Modes.NOT_EXPANDED
        assert state.get_batch_shape_mode(Normal(torch.zeros(5, 6), 1.0), B, K) == Modes.NOT_EXPANDED
    with pytest.warns(RuntimeWarning):
        assert state.get_batch_shape_mode(Normal(torch.zeros(B), 1.0), B, K) == Modes.BATCH_EXPANDED
    with pytest.warns(RuntimeWarning):
        assert state.get_batch_shape_mode(Normal(torch.zeros(B, 7), 1.0), B, K) == Modes.BATCH_EXPANDED
    with pytest.warns(RuntimeWarning):
        assert state.get_batch_shape_mode(Normal(torch.zeros(B, K, 4), 1.0), B, K) == Modes.FULLY_EXPANDED


# ---- state.sample (test/test_state.py:86-193) ----------------------------------------------------
@pytest.mark.parametrize("dims", [(), (4,), (4, 5)])
def test_sample_shapes_all_modes(dims):
    B, K = 2, 3
    cases = [(Normal(torch.zeros(*dims), 1.0) if dims else Normal(0.0, 1.0), Modes.NOT_EXPANDED),
             (Normal(torch.zeros(B, *dims), 1.0), Modes.BATCH_EXPANDED),
             (Normal(torch.zeros(B, K, *dims), 1.0), Modes.FULLY_EXPANDED)]
    for dist, mode in cases:
        state.set_batch_shape_mode(dist, mode)
        assert state.sample(dist, B, K).shape == (B, K) + dims
    nested = {"a": state.set_batch_shape_mode(Normal(torch.zeros(B), 1.0), Modes.BATCH_EXPANDED),
              "b": state.set_batch_shape_mode(Normal(0.0, 1.0), Modes.NOT_EXPANDED)}
    out = state.sample(nested, B, K)
    assert out["a"].shape == (B, K) and out["b"].shape == (B, K)
    tensor = torch.zeros(B, K)
    assert state.sample(tensor, B, K) is tensor


def test_sample_errors():
    with pytest.raises(ValueError):   # not reparameterisable (state.py:97-100)
        state.sample(torch.distributions.Categorical(torch.ones(3)), 2, 3)
    with pytest.raises(AttributeError):
        state.sample("not a distribution", 2, 3)
    with pytest.raises(ValueError):   # unsupported mode tag (state.py:93-95)
        state.sample(state.set_batch_shape_mode(Normal(0.0, 1.0), "bogus"), 2, 3)


def test_sample_batch_expanded_is_transposed_view_and_uses_k_first_noise():
    """state.py:102-103: BATCH_EXPANDED draws [K, B] noise and returns its transpose."""
    B, K = 3, 5
    dist = state.set_batch_shape_mode(Normal(torch.zeros(B), 1.0), Modes.BATCH_EXPANDED)
    with replay.record() as tape:
        x = state.sample(dist, B, K)
    assert tape.normals[0].shape == (K, B)
    assert x.shape == (B, K) and x.stride() == (1, B)
    np.testing.assert_array_equal(x.numpy(), tape.normals[0].T)


# ---- state.log_prob (test/test_state.py:196-268) -------------------------------------------------
@pytest.mark.parametrize("dims", [(), (4,), (4, 5)])
def test_log_prob_shapes_and_values(dims):
    B, K = 2, 3
    value = torch.randn(B, K, *dims)
    full_loc = torch.randn(B, K, *dims)
    batch_loc = torch.randn(B, *dims)
    want_full = Normal(full_loc, 1.5).log_prob(value).reshape(B, K, -1).sum(2)
    want_batch = Normal(batch_loc.unsqueeze(1).expand(B, K, *dims), 1.5).log_prob(value).reshape(B, K, -1).sum(2)
    want_none = Normal(torch.zeros(B, K, *dims), 1.5).log_prob(value).reshape(B, K, -1).sum(2)
    torch.testing.assert_close(state.log_prob(Normal(full_loc, 1.5), value), want_full)
    torch.testing.assert_close(state.log_prob(Normal(batch_loc, 1.5), value), want_batch)
    none = Normal(torch.zeros(*dims), 1.5) if dims else Normal(0.0, 1.5)
    torch.testing.assert_close(state.log_prob(none, value), want_none)


def test_log_prob_event_shapes_and_errors():
    B, K, C = 2, 3, 4
    one_hot = torch.distributions.OneHotCategorical(probs=torch.ones(C) / C)
    value = one_hot.sample((B, K))
    assert state.log_prob(one_hot, value).shape == (B, K)
    with pytest.raises(RuntimeError):   # state.py:146-150
        state.log_prob(Normal(torch.zeros(2, 3, 4, 5), 1.0), torch.zeros(2, 3))
    with pytest.raises(AttributeError):
        state.log_prob(3.0, torch.zeros(2, 3))
    with pytest.raises(ValueError):     # _validate_sample, state.py:142
        state.log_prob(torch.distributions.Gamma(torch.ones(2, 3), 1.0), -torch.ones(2, 3))
    both = {"a": Normal(0.0, 1.0), "b": Normal(1.0, 2.0)}
    value = {"a": torch.randn(B, K), "b": torch.randn(B, K)}
    torch.testing.assert_close(state.log_prob(both, value),
                               Normal(0.0, 1.0).log_prob(value["a"]) + Normal(1.0, 2.0).log_prob(value["b"]))


# ---- state.resample / expand_observation (test/test_state.py:272-334) ----------------------------
def test_resample_and_expand(oracle_backend):
    idx = torch.zeros(3, 2, dtype=torch.int64)
    for shape in [(3, 2), (3, 2, 4, 5)]:
        assert state.resample(torch.rand(*shape), idx).shape == shape
    got = state.resample(torch.tensor([[1.0, 2, 3], [4, 5, 6]]), torch.tensor([[1, 2, 0], [0, 0, 1]]))
    assert torch.equal(got, torch.tensor([[2.0, 3, 1], [4, 4, 5]]))
    nested = state.resample({"x": torch.rand(3, 2), "y": torch.rand(3, 2, 4)}, idx)
    assert nested["x"].shape == (3, 2) and nested["y"].shape == (3, 2, 4)
    with pytest.raises(AssertionError):
        state.resample(torch.rand(3, 2), torch.zeros(3, 5, dtype=torch.int64))
    with pytest.raises(AttributeError):
        state.resample([1, 2, 3], idx)
    for dims in [(), (4,), (4, 5)]:
        obs = torch.rand(2, *dims)
        expanded = state.expand_observation(obs, 3)
        assert expanded.shape == (2, 3) + dims and expanded.stride(1) == 0
    assert state.expand_observation({"a": torch.rand(2, 4)}, 3)["a"].shape == (2, 3, 4)


def test_resample_gradient_is_scatter_add(oracle_backend):
    value = torch.randn(2, 6, 3, dtype=torch.float64, requires_grad=True)
    idx = torch.tensor([[0, 0, 2, 2, 2, 5], [1, 1, 1, 1, 4, 4]])
    out = state.resample(value, idx)
    weights = torch.randn_like(out)
    (out * weights).sum().backward()
    reference = value.detach().clone().requires_grad_()
    (torch.gather(reference, 1, idx[..., None].expand(2, 6, 3)) * weights).sum().backward()
    torch.testing.assert_close(value.grad, reference.grad)


# ---- math (test/test_math.py) --------------------------------------------------------------------
def test_math_torch_branch(oracle_backend):
    for shape, dim in [((2, 3, 4), 0), ((2, 3, 4), 1), ((2, 3, 4), 2), ((5,), 0)]:
        x = torch.randn(*shape, dtype=torch.float64)
        torch.testing.assert_close(amath.lognormexp(x, dim=dim), x - torch.logsumexp(x, dim, keepdim=True))
        torch.testing.assert_close(amath.exponentiate_and_normalize(x, dim=dim), torch.softmax(x, dim))
    x = torch.tensor([1.0, 2.0, 3.0])
    want = torch.log(torch.exp(x) / torch.exp(x).sum())
    torch.testing.assert_close(amath.lognormexp(x), want, atol=1e-6, rtol=0)
    assert isinstance(amath.lognormexp(x), torch.Tensor)


# ---- inference: golden fixtures through the host loop --------------------------------------------
def run_infer(case, device, **flags):
    parts, named = case.build_parts(state, device)
    observations = case.observations(device)
    smc = case.meta["algorithm"] == "aesmc"
    with replay.replay(case.tape()):
        result = inference.infer("smc" if smc else "is", observations, parts["initial"],
                                 parts["transition"], parts["emission"], parts["proposal"],
                                 case.meta["num_particles"], **flags)
    return result, parts, named, observations


@pytest.mark.parametrize("name", INFER_CASES)
def test_infer_reproduces_golden_on_host(oracle_backend, name):
    case = Golden(name)
    smc = case.meta["algorithm"] == "aesmc"
    result, parts, named, observations = run_infer(
        case, torch.device("cpu"), return_log_marginal_likelihood=True, return_latents=True,
        return_original_latents=smc, return_log_weights=True, return_ancestral_indices=smc)
    tol = dict(rtol=2e-6, atol=2e-6) if case.dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
    for got, want in zip(result["log_weights"], case.series("out_log_weights")):
        np.testing.assert_allclose(got.detach().numpy(), want, **tol)
    if smc:
        for got, want in zip(result["ancestral_indices"], case.series("out_idx")):
            np.testing.assert_array_equal(got.numpy(), want)
        assert all(a.dtype == torch.int64 for a in result["ancestral_indices"])
    for got, want in zip(result["latents"], case.series("out_latents")):
        np.testing.assert_allclose(got.detach().numpy(), want, **tol)
    lml_tol = dict(rtol=1e-5, atol=1e-5) if case.dtype == torch.float32 else tol
    np.testing.assert_allclose(result["log_marginal_likelihood"].detach().numpy(), case["out_lml"], **lml_tol)
    np.testing.assert_allclose(result["last_latent"].detach().numpy(), case["out_last_latent"], **tol)

    with replay.replay(case.tape()):
        loss = losses.get_loss(observations, case.meta["num_particles"], case.meta["algorithm"],
                               parts["initial"], parts["transition"], parts["emission"], parts["proposal"])
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(case["out_loss"]), rtol=lml_tol["rtol"])
    for pname, p in named.items():
        want = case["grad_" + pname]
        scale = np.abs(want).max() + 1e-30
        np.testing.assert_allclose(p.grad.numpy() / scale, want / scale, rtol=0,
                                   atol=5e-5 if case.dtype == torch.float32 else 1e-10)


def test_infer_return_flags_and_keys(oracle_backend):
    """inference.py:187-193: seven keys always; unrequested entries are None; last_latent always."""
    case = Golden("c1_lgssm1d_smc_f32")
    result, *_ = run_infer(case, torch.device("cpu"))
    assert set(result) == {"log_marginal_likelihood", "latents", "original_latents", "log_weight",
                           "log_weights", "ancestral_indices", "last_latent"}
    assert result["log_marginal_likelihood"] is None and result["original_latents"] is None
    assert result["log_weights"] is None and result["ancestral_indices"] is None
    assert len(result["latents"]) == 8 and result["log_weight"].shape == (2, 16)
    assert result["last_latent"].shape == (2, 16)
    result, *_ = run_infer(case, torch.device("cpu"), return_latents=False, return_log_weight=False,
                           return_log_marginal_likelihood=True)
    assert result["latents"] is None and result["log_weight"] is None
    assert result["log_marginal_likelihood"].shape == (2,)


def test_infer_error_behaviour(oracle_backend):
    case = Golden("c1_lgssm1d_is_f32")
    with pytest.raises(ValueError):       # inference.py:71-74
        inference.infer("pf", [torch.zeros(2)], None, None, None, None, 4)
    with pytest.raises(RuntimeWarning):   # inference.py:169-171
        run_infer(case, torch.device("cpu"), return_original_latents=True)
    with pytest.raises(RuntimeWarning):   # inference.py:184-186
        run_infer(case, torch.device("cpu"), return_ancestral_indices=True)
    with pytest.raises(UnboundLocalError):  # losses.py:45-50
        losses.get_loss([torch.zeros(2)], 4, "vae", None, None, None, None)


def test_nan_log_weight_raises_floating_point_error(oracle_backend):
    """inference.py:244-245.  Inside infer the check is deferred to the end of the call."""
    with pytest.raises(FloatingPointError):
        inference.sample_ancestral_index(torch.tensor([[0.0, float("nan"), 1.0]]))
    model = models.LgssmNd(2, seed=0)

    def bad_emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        return state.set_batch_shape_mode(Normal(dist.loc * float("nan"), 1.0, validate_args=False),
                                          Modes.FULLY_EXPANDED)

    observations = model.simulate(3, 2, seed=0)
    with pytest.raises(FloatingPointError):
        inference.infer("smc", observations, model.initial, model.transition, bad_emission,
                        model.proposal, 8)


def test_degenerate_row_raises_like_out_of_range_gather(oracle_backend):
    """A row of all -inf log-weights: reference's digitize returns K and torch.gather raises."""
    model = models.LgssmNd(2, seed=0)

    def dead_emission(latents=None, time=None, previous_observations=None):
        dist = model.emission(latents=latents, time=time)
        return state.set_batch_shape_mode(Normal(dist.loc, 1e-30), Modes.FULLY_EXPANDED)

    observations = [1e6 * o for o in model.simulate(3, 2, seed=0)]
    with pytest.raises(RuntimeError):
        inference.infer("smc", observations, model.initial, model.transition, dead_emission,
                        model.proposal, 8)


def test_sample_ancestral_index_contract(oracle_backend):
    """test/test_inference.py:44-84: shapes, LongTensor, frequencies; plus RNG consumption."""
    for shape in [(2, 3), (1, 2), (2, 1), (1, 1)]:
        out = inference.sample_ancestral_index(torch.rand(*shape))
        assert out.shape == shape and isinstance(out, torch.LongTensor)
    weight = [0.2, 0.3, 0.5]
    trials = 10000
    idx = inference.sample_ancestral_index(torch.log(torch.tensor(weight)).unsqueeze(0).expand(trials, 3))
    freq = [(idx == i).float().sum().item() / (trials * 3) for i in range(3)]
    np.testing.assert_allclose(freq, weight, atol=1e-2)
    np.random.seed(3)
    with replay.record() as tape:
        inference.sample_ancestral_index(torch.rand(5, 4))
    assert len(tape.uniforms) == 1 and tape.uniforms[0].shape == (5, 1)  # inference.py:250


def test_get_resampled_latents_known_answer(oracle_backend):
    """test/test_inference.py:13-40 (indices there are NOT sorted: arbitrary index order works)."""
    latents = [torch.tensor([[1.0, 2, 3]]), torch.tensor([[4.0, 5, 6]]), torch.tensor([[7.0, 8, 9]]),
               torch.tensor([[10.0, 11, 12]])]
    indices = [torch.tensor([[0, 2, 1]]), torch.tensor([[2, 0, 0]]), torch.tensor([[1, 2, 0]])]
    want = [[1, 1, 2], [4, 4, 6], [8, 9, 7], [10, 11, 12]]
    for got, expected in zip(inference.get_resampled_latents(latents, indices), want):
        np.testing.assert_array_equal(got[0].numpy(), expected)
    assert len(inference.get_resampled_latents(latents[:1], [])) == 1
    with pytest.raises(AssertionError):
        inference.get_resampled_latents(latents, indices[:1])


def test_lazy_history_equals_eager_history(oracle_backend):
    """The lazy `previous_latents` must be indistinguishable from the reference's eager list for a
    model that reads the WHOLE history (non-Markov), and must gather only what is read."""
    d, B, K, T = 2, 3, 12, 5
    model = models.LgssmNd(d, seed=1, dtype=torch.float64)
    seen = []

    def transition(previous_latents=None, time=None, previous_observations=None):
        seen.append(previous_latents)
        assert len(previous_latents) == time
        mean = sum(x for x in previous_latents) / len(previous_latents)      # iteration
        mean = mean + 0.1 * previous_latents[0] + 0.0 * sum(previous_latents[-2:])  # indexing, slicing
        return state.set_batch_shape_mode(Normal(mean @ model.A.t(), 1.0), Modes.FULLY_EXPANDED)

    observations = model.simulate(T, B, seed=3)
    outs = {}
    for mode in ("lazy", "eager"):
        inference.set_history_mode(mode)
        try:
            np.random.seed(0)
            torch.manual_seed(0)
            outs[mode] = inference.infer("smc", observations, model.initial, transition, model.emission,
                                         model.proposal, K, return_log_marginal_likelihood=True,
                                         return_log_weights=True, return_ancestral_indices=True)
        finally:
            inference.set_history_mode("lazy")
    # (float64 rounding apart: in lazy mode the proposal's `previous_latents[-1] @ Wx.t() + c` is RECORDED on the lazy
    #  latent and evaluated by the fused kernels' fma chains, in eager mode by PyTorch's matmul)
    for a, b in zip(outs["lazy"]["log_weights"], outs["eager"]["log_weights"]):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
    for a, b in zip(outs["lazy"]["ancestral_indices"], outs["eager"]["ancestral_indices"]):
        assert torch.equal(a, b)
    assert isinstance(seen[0], inference.ResampledHistory) and isinstance(seen[-1], list)
    lazy = inference.ResampledHistory([torch.zeros(1, 2), torch.ones(1, 2)], torch.tensor([[1, 1]]))
    assert len(lazy) == 2 and lazy._cache == {}
    lazy[-1]
    assert list(lazy._cache) == [1]
    with pytest.raises(IndexError):
        lazy[2]
    with pytest.raises(ValueError):
        inference.set_history_mode("sometimes")


def test_observations_as_stacked_tensor(oracle_backend):
    """test/test_inference.py:176-177 passes a [T, B] tensor instead of a list."""
    case = Golden("c1_lgssm1d_smc_f64")
    parts, _ = case.build_parts(state, torch.device("cpu"))
    stacked = torch.stack(case.observations(torch.device("cpu")))
    with replay.replay(case.tape()):
        result = inference.infer("smc", stacked, parts["initial"], parts["transition"],
                                 parts["emission"], parts["proposal"], case.meta["num_particles"],
                                 return_log_marginal_likelihood=True)
    np.testing.assert_allclose(result["log_marginal_likelihood"].detach().numpy(), case["out_lml"], rtol=1e-12)


# ---- statistics (test/test_statistics.py) --------------------------------------------------------
def test_statistics(oracle_backend):
    B, K = 3, 7
    value = torch.randn(B, K, 4, dtype=torch.float64)
    log_weight = torch.randn(B, K, dtype=torch.float64)
    w = torch.softmax(log_weight, 1)
    mean = (w[..., None] * value).sum(1)
    torch.testing.assert_close(statistics.empirical_mean(value, log_weight), mean)
    torch.testing.assert_close(statistics.empirical_expectation(value, log_weight, lambda x: x), mean)
    torch.testing.assert_close(statistics.empirical_variance(value, log_weight),
                               (w[..., None] * value ** 2).sum(1) - mean ** 2)
    for offset in (0.0, 1e6, -1e6):      # test_statistics.py:71-115: stable under huge offsets
        torch.testing.assert_close(statistics.ess(torch.zeros(B, K, dtype=torch.float64) + offset),
                                   torch.full((B,), float(K), dtype=torch.float64))
    assert statistics.log_ess(torch.zeros(K, dtype=torch.float64)).shape == ()
    one_hot = torch.full((B, K), -1e30, dtype=torch.float64)
    one_hot[:, 0] = 0.0
    torch.testing.assert_close(statistics.ess(one_hot), torch.ones(B, dtype=torch.float64))


# ---- train (test/test_losses.py:11-79) -----------------------------------------------------------
def test_train_loop_runs_and_learns(oracle_backend):
    torch.manual_seed(0)
    np.random.seed(0)
    prior = models.GaussianPrior(0.0, 1.0)
    likelihood = models.GaussianLikelihood(1.0)
    network = models.GaussianInferenceNetwork(0.1, 0.0, 1.5)
    true_prior, true_likelihood = models.GaussianPrior(1.0, 1.0), models.GaussianLikelihood(0.5)
    loader = train.get_synthetic_dataloader(true_prior, None, true_likelihood, 1, 10)
    seen = []

    def callback(epoch_idx, it_idx, loss, initial, transition, emission, proposal):
        seen.append((epoch_idx, it_idx, loss.item()))
        assert initial is prior and transition is None and emission is likelihood and proposal is network

    train.train(loader, 4, "iwae", prior, None, likelihood, network, num_epochs=1,
                num_iterations_per_epoch=60, optimizer_algorithm=torch.optim.SGD,
                optimizer_kwargs={"lr": 0.05}, callback=callback)
    assert len(seen) == 60 and seen[-1][:2] == (0, 59)
    assert np.mean([s[2] for s in seen[-10:]]) < np.mean([s[2] for s in seen[:10]])
    assert train.get_chained_params(lambda: 0, None) is None
    assert len(list(train.get_chained_params(prior, None, likelihood, network))) == 5
    latents, observations = statistics.sample_from_prior(true_prior, None, true_likelihood, 1, 6)
    assert latents[0].shape == (6,) and observations[0].shape == (6,)


@pytest.mark.parametrize("name", ["train_iwae_gaussian", "train_aesmc_lgssm1d"])
def test_train_reproduces_reference_training_run(oracle_backend, name):
    """tests/golden/train_*.npz: aesmc.train.train run by the reference for 2 epochs x 2
    iterations on its own SyntheticDataset.  Same seeds here must give the same losses, the same
    final parameters and leave both random streams at the same position (i.e. data generation,
    proposal sampling and resampling consume torch's and numpy's generators in the same order,
    including the batch the reference fetches and drops at each epoch end, train.py:29-32)."""
    case = Golden(name)
    meta = case.meta
    if meta["model"] == "gaussian":
        true = (models.GaussianPrior(meta["true"][0], meta["true"][1]), None,
                models.GaussianLikelihood(meta["true"][2]))
        parts = {"initial": models.GaussianPrior(0.0, meta["prior_std"]), "transition": None,
                 "emission": models.GaussianLikelihood(1.0),
                 "proposal": models.GaussianInferenceNetwork(0.0, 0.0, 1.0)}
    else:
        true = (models.Lgssm1dInitial(*meta["initial"]),
                models.Lgssm1dTransition(meta["true"][0], meta["transition_scale"]),
                models.Lgssm1dEmission(meta["true"][1], meta["emission_scale"]))
        parts = {"initial": models.Lgssm1dInitial(*meta["initial"]),
                 "transition": models.Lgssm1dTransition(0.0, meta["transition_scale"]),
                 "emission": models.Lgssm1dEmission(0.0, meta["emission_scale"]),
                 "proposal": models.Lgssm1dProposal(*meta["proposal_scales"])}
    named = {"{}.{}".format(part, pname): p for part, module in parts.items()
             if isinstance(module, torch.nn.Module) for pname, p in module.named_parameters()}
    assert sorted(named) == meta["param_names"]
    with torch.no_grad():
        for pname, p in named.items():
            p.copy_(torch.from_numpy(case["init_" + pname]))
    torch.manual_seed(meta["seed"] + 1)
    np.random.seed(meta["seed"] + 1)
    loader = train.get_synthetic_dataloader(*true, meta["num_timesteps"], meta["batch_size"])
    seen = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        train.train(loader, meta["num_particles"], meta["algorithm"], parts["initial"], parts["transition"],
                    parts["emission"], parts["proposal"], num_epochs=2, num_iterations_per_epoch=2,
                    optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.05},
                    callback=lambda e, i, loss, *rest: seen.append(float(loss)))
    np.testing.assert_allclose(seen, case["losses"], rtol=2e-6)
    for pname, p in named.items():
        np.testing.assert_allclose(p.detach().numpy(), case["final_" + pname], rtol=2e-5, atol=1e-6)
    np.testing.assert_array_equal(torch.rand(3).numpy(), case["torch_state_probe"])
    np.testing.assert_array_equal(np.random.uniform(size=3), case["numpy_state_probe"])


def test_package_surface():
    """aesmc/__init__.py:1-7."""
    for name in ("inference", "losses", "math", "state", "statistics", "train"):
        assert hasattr(aesmc_amd, name)
    assert aesmc_amd.__version__ == "0.5.1"      # = the C ABI (tests/test_library.py holds the two together)


def test_public_surface_matches_the_reference_signature_table():
    """tests/golden/api_signatures.json (oracle/capture_golden.py: inspect.signature over the
    reference's six modules): every public function / class exists here under the same name with
    the same parameters in the same order, the same set of defaulted parameters and the same
    constant defaults; this package may only ADD trailing parameters that have defaults."""
    import enum
    import inspect
    import json
    import os
    import aesmc_amd
    from tests.golden_io import GOLDEN_DIR
    with open(os.path.join(GOLDEN_DIR, "api_signatures.json")) as fh:
        table = json.load(fh)
    assert len(table) >= 20
    for qualified, entry in table.items():
        module_name, name = qualified.split(".")
        member = getattr(getattr(aesmc_amd, module_name), name)
        if "enum" in entry:
            assert issubclass(member, enum.Enum) and sorted(m.name for m in member) == entry["enum"], qualified
            continue
        target = member.__init__ if inspect.isclass(member) else member
        mine = list(inspect.signature(target).parameters.values())
        want = entry["params"]
        assert [p.name for p in mine[:len(want)]] == [n for n, _ in want], qualified
        for p, (_, default) in zip(mine, want):
            if default is None:
                assert p.default is inspect.Parameter.empty, (qualified, p.name)
            elif default != "<object>":
                assert repr(p.default) == default, (qualified, p.name, p.default, default)
            else:
                assert p.default is not inspect.Parameter.empty, (qualified, p.name)
        assert all(p.default is not inspect.Parameter.empty for p in mine[len(want):]), qualified


# ---- linear-Gaussian callables (AffineNormal) on the oracle backend ------------------------------------
def test_affine_normal_is_a_normal_with_a_lazy_location(oracle_backend):
    from aesmc_amd.linear_gaussian import AffineNormal
    torch.manual_seed(0)
    source, weight = torch.randn(3, 7, 4, dtype=torch.float64), torch.randn(5, 4, dtype=torch.float64)
    offset, scale = torch.randn(3, 5, dtype=torch.float64), torch.tensor(0.4, dtype=torch.float64)
    dist = AffineNormal(source, weight, scale, offset=offset)
    assert isinstance(dist, torch.distributions.Normal) and dist._loc is None      # nothing evaluated yet
    assert dist.batch_shape == (3, 7, 5) and dist.event_shape == () and dist.has_rsample
    plain = torch.distributions.Normal(source @ weight.t() + offset.unsqueeze(1), scale)
    torch.testing.assert_close(dist.loc, plain.loc, rtol=1e-13, atol=1e-13)
    value = torch.randn(3, 7, 5, dtype=torch.float64)
    torch.testing.assert_close(dist.log_prob(value), plain.log_prob(value), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(dist.entropy(), plain.entropy())
    with pytest.raises(ValueError):
        AffineNormal(source, torch.randn(5, 3, dtype=torch.float64), scale)           # weight does not map source
    with pytest.raises(ValueError):
        AffineNormal(source, weight, scale, offset=torch.randn(7, 5, dtype=torch.float64))
    with pytest.raises(ValueError):
        AffineNormal(source, weight, torch.ones(2, dtype=torch.float64))
    shared = AffineNormal(source, weight, 0.4, offset=torch.zeros(5, dtype=torch.float64))
    assert shared.scale.shape == (3, 7, 5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("algorithm", ["aesmc", "iwae"])
def test_affine_callables_give_the_loss_and_gradients_of_matmul_callables(oracle_backend, dtype, algorithm):
    """The same LGSSM stated both ways — Normal(x @ W.T + c, s) and AffineNormal(x, W, s, c) — through
    get_loss + backward on the oracle backend: the fused route must be the one taken (K9 / K10 launches
    counted) and give the same numbers (float64: indices identical, loss 1e-12, gradients 1e-9)."""
    from aesmc_amd import _kernels
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    calls = {"affine_rsample": 0, "affine_logweight": 0, "affine_propagate": 0, "affine_logweight_backward": 0,
             "affine_step_backward": 0, "particle_affine_backward": 0}
    originals = {name: getattr(provider, name) for name in calls}
    for name in calls:
        def spy(*args, _name=name, **kwargs):
            calls[_name] += 1
            return originals[_name](*args, **kwargs)
        setattr(provider, name, spy)
    results = {}
    T = 5
    for affine in (False, True):
        model = LgssmNd(3, dtype=dtype, affine=affine).tune_proposal()
        observations = model.simulate(T, 4, seed=1)
        torch.manual_seed(5)
        np.random.seed(5)
        # (the matmul statement evaluated by PyTorch itself: with lazy latents its `x @ W.t() + c` would be recorded
        #  and take the very kernels it is the cross-check of)
        with inference.lazy_gather(affine):
            loss = losses.get_loss(observations, 32, algorithm, model.initial, model.transition, model.emission,
                                   model.proposal)
        loss.backward()
        results[affine] = (loss.detach(), {name: p.grad.clone() for name, p in model.named_parameters()
                                           if p.grad is not None})
    # (the model's proposal defers its draw: under SMC one launch draws and weighs, K15; importance sampling's
    # aliased transition is not such a step and K9 fills the draw)
    assert calls["affine_rsample"] + calls["affine_propagate"] == T - 1
    # importance sampling hands `transition` the list that already holds the current draw (the reference's
    # aliasing): its source is x_t, the proposal's x_{t-1} — not one
    # linear-Gaussian step, so the locations are materialised there
    fused_steps = T - 1 if algorithm == "aesmc" else 0
    # every such step's latent is the proposal's own draw: its whole backward is one K14 launch — no K12, and
    # no backward launch of the draw (K11) either
    assert calls["affine_propagate"] == fused_steps and calls["affine_logweight"] == 0
    assert calls["affine_step_backward"] == fused_steps
    if algorithm == "aesmc":
        # (the one K11 launch left is time 0's emission location, materialised for K5)
        assert calls["affine_logweight_backward"] == 0 and calls["particle_affine_backward"] == 1
    (loss_a, grads_a), (loss_b, grads_b) = results[False], results[True]
    loss_tol, grad_tol = (1e-12, 1e-9) if dtype == torch.float64 else (2e-5, 2e-3)
    assert abs(float(loss_a - loss_b)) <= loss_tol * max(1.0, abs(float(loss_a)))
    assert sorted(grads_a) == sorted(grads_b)
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= grad_tol * scale, name


@pytest.mark.parametrize("which", ["proposal", "emission"])
def test_a_detached_source_keeps_a_step_off_the_fused_route(oracle_backend, which):
    """`x` and `x.detach()` share storage but not gradients.  A model that stops the gradient into
    its proposal (or emission) by building it on the detached latent must NOT be weighed as one fused
    linear-Gaussian step (one gradient slot per operand): its gradients must equal the matmul statement's."""
    from aesmc_amd.testing.models import LgssmNd

    class Detached(LgssmNd):
        def proposal(self, previous_latents=None, time=None, observations=None):
            if time > 0 and which == "proposal":
                previous_latents = [previous_latents[-1].detach()]
            return super().proposal(previous_latents=previous_latents, time=time, observations=observations)

        def emission(self, latents=None, time=None, previous_observations=None):
            if which == "emission":
                latents = [latents[-1].detach()]
            return super().emission(latents=latents, time=time, previous_observations=previous_observations)

    results = {}
    for affine in (False, True):
        model = Detached(3, dtype=torch.float64, affine=affine, defer_draw=False).tune_proposal()
        observations = model.simulate(4, 3, seed=1)
        torch.manual_seed(7)
        np.random.seed(7)
        with inference.lazy_gather(affine):
            loss = losses.get_loss(observations, 16, "aesmc", model.initial, model.transition, model.emission,
                                   model.proposal)
        loss.backward()
        results[affine] = (loss.detach(), {name: p.grad.clone() for name, p in model.named_parameters()
                                           if p.grad is not None})
    (loss_a, grads_a), (loss_b, grads_b) = results[False], results[True]
    assert abs(float(loss_a - loss_b)) <= 1e-12 * max(1.0, abs(float(loss_a)))
    assert sorted(grads_a) == sorted(grads_b)
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= 1e-9 * scale, name


def test_affine_normal_outside_the_fused_route_materialises_its_location(oracle_backend):
    """A step whose three terms are not all AffineNormal in the right tensors (here: the emission reads
    a COPY of the latent) takes the ordinary route: `.loc` is evaluated and the numbers are the same.
    Such an emission READS the newest latent's values, so its model must not set `defer_draw`."""
    from aesmc_amd.linear_gaussian import AffineNormal
    from aesmc_amd.testing.models import LgssmNd

    class CopyingEmission(LgssmNd):
        def emission(self, latents=None, time=None, previous_observations=None):
            return self._tag(AffineNormal(latents[-1].clone(), self.C, self.emission_scale), "FULLY_EXPANDED")

    outs = []
    for cls in (LgssmNd, CopyingEmission):
        model = cls(2, dtype=torch.float64, affine=True, defer_draw=False)
        observations = model.simulate(4, 3, seed=2)
        torch.manual_seed(1)
        np.random.seed(1)
        outs.append(inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                    model.proposal, 16, return_log_marginal_likelihood=True,
                                    return_ancestral_indices=True))
    for a, b in zip(outs[0]["ancestral_indices"], outs[1]["ancestral_indices"]):
        assert torch.equal(a, b)
    torch.testing.assert_close(outs[0]["log_marginal_likelihood"], outs[1]["log_marginal_likelihood"],
                               rtol=1e-12, atol=1e-12)


def test_fused_nonlinear_model_equals_the_plain_one(oracle_backend):
    """BASELINE.json's nonlinear model with `fused=True` (d x d maps through particle_affine / AffineNormal,
    K8, here on the oracle backend; the proposal net stays PyTorch's) against the same
    model with PyTorch matmuls: float64 loss and every gradient."""
    from aesmc_amd.testing.models import NonlinearSsm
    results = []
    for fused in (False, True):
        model = NonlinearSsm(3, hidden=12, dtype=torch.float64, fused=fused)
        observations = model.simulate(4, 3, seed=2)
        torch.manual_seed(9)
        np.random.seed(9)
        loss = losses.get_loss(observations, 24, "aesmc", model.initial, model.transition, model.emission,
                               model.proposal)
        loss.backward()
        results.append((loss.detach(), {name: p.grad.clone() for name, p in model.named_parameters()
                                        if p.grad is not None}))
    (loss_a, grads_a), (loss_b, grads_b) = results
    assert abs(float(loss_a - loss_b)) <= 1e-12 * max(1.0, abs(float(loss_a)))
    assert sorted(grads_a) == sorted(grads_b)
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= 1e-9 * scale, name


@pytest.mark.parametrize("name", ["lgssm3d_smc_f64", "lgssm10d_smc_f64"])
def test_affine_route_reproduces_the_reference_fixtures_on_host(oracle_backend, name):
    """The reference-captured LGSSM fixtures through the model stated with AffineNormal callables (the C
    oracle standing in for kernels K8 - K12): every ancestor index of the reference, log-weights, log Z,
    loss and parameter gradients — the linear-Gaussian route pinned to the reference's own outputs."""
    case = Golden(name)
    parts, named = case.build_parts(state, torch.device("cpu"), affine=True)
    observations = case.observations(torch.device("cpu"))
    with replay.replay(case.tape()):
        result = inference.infer("smc", observations, parts["initial"], parts["transition"], parts["emission"],
                                 parts["proposal"], case.meta["num_particles"], return_log_marginal_likelihood=True,
                                 return_latents=False, return_log_weights=True, return_ancestral_indices=True)
    for got, want in zip(result["ancestral_indices"], case.series("out_idx")):
        np.testing.assert_array_equal(got.numpy(), want)
    for got, want in zip(result["log_weights"], case.series("out_log_weights")):
        np.testing.assert_allclose(got.detach().numpy(), want, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(result["log_marginal_likelihood"].detach().numpy(), case["out_lml"], rtol=1e-10, atol=1e-10)
    with replay.replay(case.tape()):
        loss = losses.get_loss(observations, case.meta["num_particles"], "aesmc", parts["initial"],
                               parts["transition"], parts["emission"], parts["proposal"])
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(case["out_loss"]), rtol=1e-10)
    for pname, p in named.items():
        want = case["grad_" + pname]
        scale = np.abs(want).max() + 1e-30
        np.testing.assert_allclose(p.grad.numpy() / scale, want / scale, rtol=0, atol=1e-9)


def test_a_forward_pass_leaves_no_reference_cycles_behind(oracle_backend):
    """An ELBO that is evaluated with autograd on and then dropped (no backward) must free its graph by
    reference counting alone: a cycle through a step node would keep every step's [B,K,d] tensors alive
    until the cyclic collector happens to run."""
    import gc
    from aesmc_amd import _ops
    from aesmc_amd.testing.models import LgssmNd
    model = LgssmNd(3, dtype=torch.float64, affine=True).tune_proposal()
    observations = model.simulate(4, 3, seed=1)
    gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    try:
        loss = losses.get_loss(observations, 16, "aesmc", model.initial, model.transition, model.emission,
                               model.proposal)
        assert loss.requires_grad
        del loss
        gc.collect()
        leaked = [o for o in gc.garbage if isinstance(o, (_ops.PendingStep, torch.Tensor))]
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
    assert not leaked, leaked


@pytest.mark.parametrize("algorithm,grad", [("smc", False), ("smc", True), ("is", False)])
def test_a_deferred_draw_gives_the_very_same_run(oracle_backend, algorithm, grad):
    """AffineNormal(..., defer_draw=True) on the proposal: the draw is produced by the launch that weighs
    the step (K15) — or by K9 when the step is not weighed that way (importance sampling's aliased
    transition; log-weights that need their own autograd node) — and every number of the run is the one
    the immediate draw gives: latents, log-weights, ancestors, evidence, gradients, RNG consumption."""
    from aesmc_amd import _kernels
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    calls = {"affine_propagate": 0, "affine_rsample": 0}
    originals = {name: getattr(provider, name) for name in calls}
    for name in calls:
        def spy(*args, _name=name, **kwargs):
            calls[_name] += 1
            return originals[_name](*args, **kwargs)
        setattr(provider, name, spy)
    T, runs = 4, {}
    try:
        for defer in (False, True):
            for name in calls:
                calls[name] = 0
            model = LgssmNd(3, dtype=torch.float64, affine=True, defer_draw=defer).tune_proposal()
            observations = model.simulate(T, 4, seed=3)
            torch.manual_seed(11)
            np.random.seed(11)
            with torch.set_grad_enabled(grad):
                out = inference.infer(algorithm, observations, model.initial, model.transition, model.emission,
                                      model.proposal, 24, return_log_marginal_likelihood=True,
                                      return_latents=True, return_log_weight=not grad,
                                      return_ancestral_indices=algorithm == "smc")
            if grad:
                (-out["log_marginal_likelihood"].mean()).backward()
            after = (torch.rand(1).item(), np.random.uniform())       # where the two RNG streams stand afterwards
            runs[defer] = (out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                           after, dict(calls))
    finally:
        for name, fn in originals.items():
            setattr(provider, name, fn)
    (a, grads_a, rng_a, calls_a), (b, grads_b, rng_b, calls_b) = runs[False], runs[True]
    assert calls_a == {"affine_propagate": 0, "affine_rsample": T - 1}
    if algorithm == "smc":      # every step from the second on: one launch draws and weighs
        assert calls_b == {"affine_propagate": T - 1, "affine_rsample": 0}
    else:                       # not a linear-Gaussian step in the proposal's x_{t-1}: K9 fills the draw
        assert calls_b == {"affine_propagate": 0, "affine_rsample": T - 1}
    assert rng_a == rng_b
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    for x, y in zip(a["latents"], b["latents"]):
        assert torch.equal(x, y)
    assert torch.equal(a["last_latent"], b["last_latent"])
    if not grad:
        assert torch.equal(a["log_weight"], b["log_weight"])
    if algorithm == "smc":
        for x, y in zip(a["ancestral_indices"], b["ancestral_indices"]):
            assert torch.equal(x, y)
    assert sorted(grads_a) == sorted(grads_b) and (not grad or grads_a)
    for name in grads_a:
        assert torch.equal(grads_a[name], grads_b[name]), name


def test_a_deferred_draw_outside_infer_is_drawn_at_once(oracle_backend):
    """`state.sample` honours `defer_draw` only inside `infer` (which guarantees the values are filled
    before anything reads them); a direct call draws immediately."""
    from aesmc_amd.linear_gaussian import AffineNormal
    source = torch.randn(3, 7, 4, dtype=torch.float64)
    weight = torch.randn(4, 4, dtype=torch.float64)
    scale = torch.tensor(0.5, dtype=torch.float64)
    torch.manual_seed(2)
    now = state.sample(AffineNormal(source, weight, scale, defer_draw=True), 3, 7)
    torch.manual_seed(2)
    ref = state.sample(AffineNormal(source, weight, scale), 3, 7)
    assert not hasattr(now, "_aesmc_pending_noise") and torch.equal(now, ref)


def test_a_callable_that_reads_a_deferred_draw_gets_its_values(oracle_backend):
    """Inside `infer` the proposal's draw is a `LazyDraw` without values until the launch that weighs the step forms
    it.  No promise is asked of the model: a callable that READS the newest latent (here: an emission that copies
    it) gets the draw formed on the spot (K9), and the run equals the one that draws at once — every number."""
    from aesmc_amd.linear_gaussian import AffineNormal
    from aesmc_amd.testing.models import LgssmNd

    class CopyingEmission(LgssmNd):
        def emission(self, latents=None, time=None, previous_observations=None):
            return self._tag(AffineNormal(latents[-1].clone(), self.C, self.emission_scale), "FULLY_EXPANDED")

    outs = {}
    for defer in (False, True):
        model = CopyingEmission(2, dtype=torch.float64, affine=True, defer_draw=defer)
        observations = model.simulate(3, 3, seed=2)
        torch.manual_seed(3)
        np.random.seed(3)
        outs[defer] = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                      model.proposal, 16, return_log_marginal_likelihood=True,
                                      return_ancestral_indices=True)
    assert torch.equal(outs[False]["log_marginal_likelihood"], outs[True]["log_marginal_likelihood"])
    for a, b in zip(outs[False]["latents"] + outs[False]["ancestral_indices"],
                    outs[True]["latents"] + outs[True]["ancestral_indices"]):
        assert torch.equal(a, b)


def test_a_reference_style_model_reaches_the_fused_route_unedited(oracle_backend):
    """The host logic of VERDICT r02 item 5 on the oracle backend: `Normal(previous_latents[-1] @ W.t() + c, s)`
    callables (LgssmNd(affine=False): the reference's own style) are recorded on the lazy latents and weighed as
    linear-Gaussian steps — launches counted — and agree with PyTorch's own evaluation of the same model."""
    from aesmc_amd import _kernels, _lazy
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    T, results = 5, {}
    for lazy in (False, True):
        calls = {"affine_propagate": 0, "affine_step_backward": 0}       # (the oracle provider gathers inside these)
        originals = {name: getattr(provider, name) for name in calls}
        for name in calls:
            def spy(*args, _name=name, **kwargs):
                calls[_name] += 1
                return originals[_name](*args, **kwargs)
            setattr(provider, name, spy)
        try:
            model = LgssmNd(3, dtype=torch.float64, affine=False).tune_proposal()
            observations = model.simulate(T, 4, seed=1)
            torch.manual_seed(5)
            np.random.seed(5)
            with inference.lazy_gather(lazy):
                loss = losses.get_loss(observations, 32, "aesmc", model.initial, model.transition, model.emission,
                                       model.proposal)
            loss.backward()
        finally:
            for name, fn in originals.items():
                setattr(provider, name, fn)
        results[lazy] = (loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                         calls)
    (loss_a, grads_a, calls_a), (loss_b, grads_b, calls_b) = results[False], results[True]
    assert calls_a["affine_propagate"] == 0 and calls_a["affine_step_backward"] == 0
    assert calls_b == {"affine_propagate": T - 1, "affine_step_backward": T - 1}
    assert abs(float(loss_a - loss_b)) <= 1e-12 * max(1.0, abs(float(loss_a)))
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= 1e-9 * scale, name
    # what is and is not recorded
    source = torch.randn(2, 5, 3, dtype=torch.float64)
    index = torch.tensor([[0, 0, 2, 3, 4], [1, 1, 1, 2, 4]])
    weight, bias, row = torch.randn(4, 3, dtype=torch.float64), torch.randn(4, dtype=torch.float64), \
        torch.randn(2, 4, dtype=torch.float64)
    moved = torch.gather(source, 1, index.unsqueeze(-1).expand_as(source))
    x = _lazy.LazyResampled(source, index)
    loc = x @ weight.t() + row.unsqueeze(1) + bias
    assert type(loc) is _lazy.LazyAffine and x.is_pending and loc.is_pending
    torch.testing.assert_close(loc.materialise(), moved @ weight.t() + row.unsqueeze(1) + bias, rtol=1e-13, atol=1e-13)
    assert type(torch.nn.functional.linear(_lazy.LazyResampled(source, index), weight, bias)) is _lazy.LazyAffine
    assert type(0.5 * _lazy.LazyResampled(source, index)) is _lazy.LazyAffine
    y = _lazy.LazyResampled(source, index)
    out = torch.tanh(y)                         # anything else: the values, then the operator
    assert not isinstance(out, _lazy.LazyParticles) and not y.is_pending and torch.equal(out, torch.tanh(moved))
    dist = torch.distributions.Normal(_lazy.LazyResampled(source, index) @ weight.t(), torch.tensor(0.5, dtype=torch.float64))
    assert type(dist.loc) is _lazy.LazyAffine and dist.loc.is_pending and dist.batch_shape == (2, 5, 4)
    # a first reader under no_grad (a diagnostic look inside a callable) does not cost later readers their gradient
    leaf = source.clone().requires_grad_(True)
    z = _lazy.LazyResampled(leaf, index)
    with torch.no_grad():
        peek = torch.tanh(z)
    assert not peek.requires_grad and not z.is_pending
    (z.materialise() * 2.0).sum().backward()
    want = torch.zeros_like(source).scatter_add_(1, index.unsqueeze(-1).expand_as(source), torch.full_like(source, 2.0))
    torch.testing.assert_close(leaf.grad, want, rtol=0, atol=0)
    # ... but a lazy MADE and read under no_grad (a pure evaluation) carries no history, as the reference's eager tensors
    # do not: nothing is retained and nothing cascades into the later steps
    parameter = torch.nn.Parameter(weight.clone())
    with torch.no_grad():
        made = _lazy.LazyResampled(leaf, index)
        located = made @ parameter.t()
        assert type(located) is _lazy.LazyAffine and not made.requires_grad and not located.requires_grad
        values = located.materialise()
        gathered = made.materialise()
    assert not values.requires_grad and values.grad_fn is None
    assert not gathered.requires_grad and gathered.grad_fn is None
    torch.testing.assert_close(values, moved @ weight.t(), rtol=1e-13, atol=1e-13)


def test_linked_step_nodes_give_the_gradients_of_unlinked_ones(oracle_backend):
    """Host logic of the folded gather backward (`_ops.StepLink`) on the oracle backend: with consecutive step nodes
    handing the per-child gradient on, the loss and every gradient equal those of the run whose every step sums
    children into ancestors itself; the linked run asks for one stand-alone sum (x_0's) instead of T - 1; handing the
    latents to the caller switches the linking off."""
    from aesmc_amd import _kernels
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    T, results = 6, {}
    for fold in (False, True):
        calls = {"gather_backward": 0, "with_children": 0}
        real_gb, real_sb = provider.gather_backward, provider.affine_step_backward

        def gb_spy(*args, **kwargs):
            calls["gather_backward"] += 1
            return real_gb(*args, **kwargs)

        def sb_spy(*args, **kwargs):
            calls["with_children"] += kwargs.get("child_grad") is not None
            return real_sb(*args, **kwargs)

        provider.gather_backward, provider.affine_step_backward = gb_spy, sb_spy
        try:
            model = LgssmNd(3, dtype=torch.float64, affine=True).tune_proposal()
            observations = model.simulate(T, 4, seed=1)
            torch.manual_seed(5)
            np.random.seed(5)
            with inference.fold_gather_backward(fold):
                loss = losses.get_loss(observations, 48, "aesmc", model.initial, model.transition, model.emission,
                                       model.proposal)
                loss.backward()
                grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
                counted = dict(calls)
                torch.manual_seed(5)
                np.random.seed(5)
                kept = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                       model.proposal, 48, return_log_marginal_likelihood=True, return_latents=False,
                                       return_log_weight=False, return_original_latents=True)
                before = calls["with_children"]
                kept["log_marginal_likelihood"].sum().backward()
                assert calls["with_children"] == before
        finally:
            provider.gather_backward, provider.affine_step_backward = real_gb, real_sb
        results[fold] = (loss.detach(), grads, counted)
    (loss_a, grads_a, calls_a), (loss_b, grads_b, calls_b) = results[False], results[True]
    assert torch.equal(loss_a, loss_b)
    assert calls_a == {"gather_backward": T - 1, "with_children": 0}
    assert calls_b == {"gather_backward": 1, "with_children": T - 2}
    assert sorted(grads_a) == sorted(grads_b) and grads_a
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= 1e-10 * scale, name


def test_shared_parameters_are_finished_once_per_run_of_steps(oracle_backend):
    """Host logic of the chained weight gradients (`_ops.StepLink.carry`): steps that receive the same A, C, Q and
    scales leave their sums to the step before, the run's first step returns the total — the same gradients as when
    every step returns its own; a model whose steps do NOT share a parameter keeps every step's own."""
    from aesmc_amd import _kernels, _ops
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    T, results = 6, {}
    for chained in (False, True):
        seen = []
        real = provider.affine_step_backward

        def spy(*args, **kwargs):
            chain = kwargs.get("chain")
            seen.append(None if chain is None else (chain["carry"] is not None, bool(chain["defer"])))
            return real(*args, **kwargs)

        provider.affine_step_backward = spy
        previous, _ops._CHAIN_SHARED = _ops._CHAIN_SHARED, chained
        try:
            model = LgssmNd(3, dtype=torch.float64, affine=True).tune_proposal()
            observations = model.simulate(T, 4, seed=1)
            torch.manual_seed(5)
            np.random.seed(5)
            loss = losses.get_loss(observations, 48, "aesmc", model.initial, model.transition, model.emission,
                                   model.proposal)
            loss.backward()
            results[chained] = ({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, list(seen))
        finally:
            provider.affine_step_backward, _ops._CHAIN_SHARED = real, previous
    (plain, seen_plain), (chained, seen_chained) = results[False], results[True]
    assert seen_plain == [None] * (T - 1)
    # backward order: the last step defers, the middle ones carry and defer, the first carries and finishes
    assert seen_chained == [(False, True)] + [(True, True)] * (T - 3) + [(True, False)]
    assert sorted(plain) == sorted(chained) and plain
    for name in plain:
        scale = max(float(plain[name].abs().max()), 1e-30)
        assert float((plain[name] - chained[name]).abs().max()) <= 1e-10 * scale, name
    # two tensors that only LOOK alike are two parameters
    a, b = torch.zeros(3, 3, requires_grad=True), torch.zeros(3, 3, requires_grad=True)
    assert _ops._same_parameter(a, a) and not _ops._same_parameter(a, b)
    assert _ops._same_parameter(a.t(), a.t()) and not _ops._same_parameter(a.t(), b.t())
    assert not _ops._same_parameter(a.t(), a.detach().t())


def test_steps_with_their_own_parameters_keep_their_own_gradients(oracle_backend):
    """A time-inhomogeneous model (a transition matrix per timestep): consecutive steps share C, Q and the scales but
    not A, so no step may leave its sums to its neighbour — every A_t gets its own gradient, equal to the unlinked
    run's; two steps in the middle that DO share every parameter chain between themselves only."""
    from aesmc_amd import _kernels, _ops
    from aesmc_amd.testing.models import LgssmNd

    class PerStep(LgssmNd):
        def __init__(self, T, shared_pair):
            super().__init__(3, dtype=torch.float64, affine=True)
            mats = [torch.nn.Parameter(self.A.detach().clone() * (1.0 - 0.01 * t)) for t in range(T)]
            if shared_pair:
                mats[3] = mats[2]
            self.As = torch.nn.ParameterList(mats)

        def transition(self, previous_latents=None, time=None, previous_observations=None):
            return self._tag(self._affine_normal(previous_latents[-1], self.As[time], self.transition_scale),
                             "FULLY_EXPANDED")

    provider = _kernels.get()
    T = 6
    for shared_pair in (False, True):
        results = {}
        for fold in (False, True):
            seen = []
            real = provider.affine_step_backward

            def spy(*args, **kwargs):
                chain = kwargs.get("chain")
                seen.append(None if chain is None else (chain["carry"] is not None, bool(chain["defer"])))
                return real(*args, **kwargs)

            provider.affine_step_backward = spy
            try:
                torch.manual_seed(0)
                model = PerStep(T, shared_pair).tune_proposal()
                observations = model.simulate(T, 4, seed=1)
                torch.manual_seed(5)
                np.random.seed(5)
                with inference.fold_gather_backward(fold):
                    loss = losses.get_loss(observations, 48, "aesmc", model.initial, model.transition, model.emission,
                                           model.proposal)
                    loss.backward()
            finally:
                provider.affine_step_backward = real
            results[fold] = ({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, seen)
        (plain, _), (linked, seen) = results[False], results[True]
        if shared_pair:      # backward order t = 5, 4, 3, 2, 1: only step 3 may leave its sums (to step 2)
            assert seen == [None, None, (False, True), (True, False), None]
        else:
            assert seen == [None] * (T - 1)
        assert sorted(plain) == sorted(linked) and any(name.startswith("As.") for name in plain)
        for name in plain:
            scale = max(float(plain[name].abs().max()), 1e-30)
            assert float((plain[name] - linked[name]).abs().max()) <= 1e-10 * scale, name


def test_recorded_locations_name_the_models_own_tensors():
    """`x @ W.t()` reaches the recording as a new view per call; what is kept is W itself when the view is all of it
    (the same object every timestep: the per-step caches and the chained weight gradients compare by identity) — and a
    view that is NOT all of W, or that differs from it in autograd's eyes, stays the view."""
    from aesmc_amd import _lazy
    W = torch.nn.Parameter(torch.randn(3, 3))
    assert _lazy._own_tensor(W.t().t()) is W
    assert _lazy._own_tensor(W.t()) is not W and _lazy._own_tensor(W[:2].t().t()) is not W
    with torch.no_grad():
        cut = W.t()
    assert _lazy._own_tensor(cut.t()) is not W          # no gradient may reach W through it
    plain = torch.randn(3, 3)
    assert _lazy._own_tensor(plain) is plain
    # an expanded one-value scale: the value's own tensor
    from aesmc_amd import linear_gaussian
    scale = torch.tensor(0.7)
    x = _lazy.LazyResampled(torch.randn(2, 5, 3), torch.zeros(2, 5, dtype=torch.int64))
    dist = torch.distributions.Normal(x @ W.t(), scale, validate_args=False)
    terms = linear_gaussian.affine_terms(dist)
    assert terms is not None and terms.weight is W and terms.scale_param is scale
    learned = torch.nn.Parameter(torch.tensor(0.7))
    terms = linear_gaussian.affine_terms(torch.distributions.Normal(x @ W.t(), learned, validate_args=False))
    assert terms.scale_param is learned
    with torch.no_grad():
        cut_off = learned.expand(2, 5, 3)
    fake = torch.distributions.Normal(x @ W.t(), 1.0, validate_args=False)
    fake.scale = cut_off             # an expanded view autograd does not connect to the parameter: stays a view
    assert linear_gaussian.affine_terms(fake).scale_param is not learned


def test_settings_are_scoped_by_context_and_keep_the_setters_meaning():
    """`settings`: the module-level setters change the process-wide defaults; `override` (what `inference.lazy_gather`,
    `inference.fold_gather_backward` and the hipGraph capture use) changes a copy for the duration of a block in THIS
    context only — another thread keeps seeing the defaults; invalid values are refused before anything changes."""
    import threading
    import pytest
    from aesmc_amd import inference, settings, state
    assert settings.current().history_mode == "lazy" and settings.current().fused_normal
    seen = {}
    with settings.override(fused_normal=False, history_mode="eager"):
        assert not settings.current().fused_normal and settings.current().history_mode == "eager"
        with inference.lazy_gather(False):
            assert not settings.current().lazy_gather and not settings.current().fused_normal
            worker = threading.Thread(target=lambda: seen.update(other=settings.current()))
            worker.start()
            worker.join()
        assert settings.current().lazy_gather
    assert seen["other"].fused_normal and seen["other"].lazy_gather and seen["other"].history_mode == "lazy"
    assert settings.current().fused_normal and settings.current().history_mode == "lazy"
    state.set_fused_normal(False)
    try:
        assert not settings.current().fused_normal
        with settings.override(kernel_noise=False):
            assert not settings.current().fused_normal and not settings.current().kernel_noise
    finally:
        state.set_fused_normal(True)
    with pytest.raises(ValueError):
        inference.set_history_mode("sometimes")
    with pytest.raises(ValueError):
        state.set_validation_mode("later")
    with pytest.raises(ValueError):
        with settings.override(validation_mode="never"):
            pass
    assert settings.current().history_mode == "lazy" and settings.current().validation_mode == "deferred"


def test_the_wide_steps_recomputing_backward_is_float64_autograds():
    """`_kernels.affine_step_backward_wide` (the backward of a step on rows wider than the fused kernels take: recomputed
    from x_{t-1}, the ancestors, x_t and the log-weights) is plain tensor algebra around the gather: on CPU float64
    tensors, with `torch.gather` standing in for K3, every one of its eleven gradients equals autograd of the step
    written with `torch.distributions` (aesmc/state.py:114-155, :179, aesmc/inference.py:108-130) to 1e-12 — offsets per
    batch row, shared and absent, the observation, the three scales, with and without a gradient arriving at x_t."""
    from aesmc_amd import _kernels
    provider = _kernels.HipKernels.__new__(_kernels.HipKernels)      # (no library needed: the method launches nothing itself)
    provider.gather = lambda src, idx: torch.gather(src, 1, idx.unsqueeze(-1).expand_as(src))
    torch.manual_seed(0)
    B, K, d = 3, 8192, 6
    make = lambda *shape: torch.randn(*shape, dtype=torch.float64)
    x_prev, eps, y, glse = make(B, K, d), make(B, K, d), make(B, d), make(B)
    ancestors = torch.sort(torch.randint(0, K, (B, K)), 1)[0]
    A, C, Q = (0.3 * make(d, d) for _ in range(3))
    scales = [torch.tensor(v, dtype=torch.float64) for v in (1.0, 0.5, 0.7)]
    leaf = lambda t: None if t is None else t.clone().requires_grad_(True)
    for off_p, off_g, off_q, gx in ((make(d), make(B, d), make(B, d), 1e-2 * make(B, K, d)), (None, None, make(d), None)):
        moved = leaf(provider.gather(x_prev, ancestors))
        A_, C_, Q_, y_, op_, og_, oq_ = (leaf(t) for t in (A, C, Q, y, off_p, off_g, off_q))
        s_ = [leaf(v) for v in scales]
        row = lambda off: 0 if off is None else (off.unsqueeze(1) if off.dim() == 2 else off)
        loc_q = moved @ Q_.t() + row(oq_)
        x_t = loc_q + s_[2] * eps
        logn = lambda v, loc, scale: torch.distributions.Normal(loc, scale).log_prob(v).sum(2)
        lw = logn(x_t, moved @ A_.t() + row(op_), s_[0]) + logn(y_.unsqueeze(1), x_t @ C_.t() + row(og_), s_[1]) - \
            logn(x_t, loc_q, s_[2])
        lse = torch.logsumexp(lw, 1)
        ((glse * lse).sum() + (0 if gx is None else (gx * x_t).sum())).backward()
        need = [True, False, True, True, off_p is not None, True, off_g is not None, True, True, True, True, True]
        got = provider.affine_step_backward_wide(x_prev, x_t.detach(), y, (A, off_p), (C, off_g), (Q, off_q), scales, need,
                                                 lw.detach(), lse.detach(), grad_lse=glse, grad_x=gx, ancestors=ancestors)
        want = {0: moved.grad, 2: y_.grad, 3: A_.grad, 5: C_.grad, 7: Q_.grad, 8: oq_.grad, 9: s_[0].grad, 10: s_[1].grad,
                11: s_[2].grad}
        if off_p is not None:
            want[4], want[6] = op_.grad, og_.grad
        for slot, reference in want.items():
            error = float((got[slot].reshape(reference.shape) - reference).abs().max())
            assert error <= 1e-12 * (1 + float(reference.abs().max())), (slot, error)
        assert got[1] is None and (off_p is not None or (got[4] is None and got[6] is None))


# ---- torch.distributions made sync-free inside `infer` (aesmc_amd/_syncfree.py) -----------------------------------
def test_syncfree_wrappers_are_stock_outside_a_scope_and_for_host_tensors():
    """The wrappers installed by the first scope delegate to PyTorch's own functions outside a scope, and inside one for
    anything that does not live on a HIP device: same objects, same values, same errors at the same place."""
    from aesmc_amd import _syncfree
    loc = torch.zeros(3)
    with _syncfree.scope():
        inside = torch.distributions.Normal(loc, 0.7)
        assert _syncfree.active()
        assert inside._validate_args is True
        with pytest.raises(ValueError, match="Expected parameter scale"):
            torch.distributions.Normal(loc, -1.0)          # a host tensor is checked on the host, at once
        with pytest.raises(ValueError, match="within the support"):
            torch.distributions.Exponential(torch.ones(3)).log_prob(-torch.ones(3))
        assert torch.distributions.Normal(loc, -1.0, validate_args=False).scale[0] == -1.0
    assert not _syncfree.active()
    outside = torch.distributions.Normal(loc, 0.7)
    torch.testing.assert_close(inside.scale, outside.scale, rtol=0, atol=0)
    torch.testing.assert_close(inside.log_prob(loc + 0.3), outside.log_prob(loc + 0.3), rtol=0, atol=0)
    with pytest.raises(ValueError, match="Expected parameter scale"):
        torch.distributions.Normal(loc, -1.0)
    # eager validation mode: PyTorch's own __init__ / _validate_sample even inside a scope
    from aesmc_amd import settings
    with _syncfree.scope(), settings.override(validation_mode="eager"):
        with pytest.raises(ValueError, match="Expected parameter scale"):
            torch.distributions.Normal(loc, -1.0)


def test_syncfree_constant_cache_keys_on_type_value_dtype_and_sign():
    from aesmc_amd import _syncfree
    cpu = torch.device("cpu")
    a = _syncfree.constant(0.7, torch.float32, cpu)
    assert a is _syncfree.constant(0.7, torch.float32, cpu)
    assert a is not _syncfree.constant(0.7, torch.float64, cpu)
    assert float(a) == float(torch.tensor(0.7, dtype=torch.float32))
    assert _syncfree.constant(1, torch.float32, cpu) is not _syncfree.constant(1.0, torch.float32, cpu)
    assert _syncfree.constant(True, torch.float32, cpu) is not _syncfree.constant(1, torch.float32, cpu)
    assert np.signbit(float(_syncfree.constant(-0.0, torch.float32, cpu)))
    assert not np.signbit(float(_syncfree.constant(0.0, torch.float32, cpu)))


def test_reference_style_1d_model_under_the_oracle_provider_matches_the_port(oracle_backend):
    """The literal reference classes (Python-number scales, default validate_args) through `infer` with the scope
    active: the CPU port of the reference on the same draws gives the same numbers."""
    from aesmc_amd.testing import replay
    from oracle import reference_port

    def parts(which):
        return (models.Lgssm1dInitial(0.0, 1.0), models.Lgssm1dTransition(0.7, 0.5, state=which),
                models.Lgssm1dEmission(0.9, 0.4, state=which), models.Lgssm1dProposal(0.8, 0.8, state=which))
    torch.manual_seed(3)
    observations = [torch.randn(4) for _ in range(5)]
    flags = dict(return_log_marginal_likelihood=True, return_log_weights=True, return_ancestral_indices=True)
    torch.manual_seed(0)
    theirs = parts(reference_port)
    np.random.seed(1)
    with replay.record() as tape:
        want = reference_port.infer("smc", observations, *theirs, 16, **flags)
    torch.manual_seed(0)
    ours = parts(state)
    with replay.replay(tape):
        got = inference.infer("smc", observations, *ours, 16, **flags)
    for a, b in zip(got["ancestral_indices"], want["ancestral_indices"]):
        assert torch.equal(a, b)
    torch.testing.assert_close(got["log_marginal_likelihood"], want["log_marginal_likelihood"], rtol=1e-5, atol=1e-5)


def test_settings_override_pins_only_the_fields_it_names():
    """`override` used to snapshot every field, so a `set_default(...)` (or a module-level setter) called
    inside any `with override(...)` block had no effect until the block ended.  Now the scoped layer holds only the
    changed fields and every read resolves scoped-else-default."""
    from aesmc_amd import settings
    before = settings.current().fused_normal
    try:
        with settings.override(lazy_gather=False):
            assert settings.current().lazy_gather is False
            state.set_fused_normal(not before)                       # a field the block did not pin: effective at once
            assert settings.current().fused_normal is (not before)
            with settings.override(fused_normal=before):             # nested: pins it for the inner block only
                assert settings.current().fused_normal is before and settings.current().lazy_gather is False
                inference.set_lazy_gather(True)                      # pinned by the outer block: the default moves, the view not
                assert settings.current().lazy_gather is False
            assert settings.current().fused_normal is (not before)
        assert settings.current().lazy_gather is True
        with pytest.raises(TypeError):
            with settings.override(no_such_field=1):
                pass
        with pytest.raises(ValueError):
            with settings.override(history_mode="sometimes"):
                pass
        with pytest.raises(AttributeError):
            with settings.override(lazy_gather=False):
                settings.current().lazy_gather = True
    finally:
        settings.set_default(fused_normal=before, lazy_gather=True)


def test_measurement_knobs_are_read_only_beside_their_switch(monkeypatch):
    from aesmc_amd import settings
    monkeypatch.setenv("AESMC_K16_PAIRS", "0")
    monkeypatch.delenv("AESMC_MEASUREMENT_KNOBS", raising=False)
    assert settings.knob("AESMC_K16_PAIRS", "1") == "1"
    monkeypatch.setenv("AESMC_MEASUREMENT_KNOBS", "1")
    assert settings.knob("AESMC_K16_PAIRS", "1") == "0"


def test_the_proposal_net_operator_and_its_backward_on_the_oracle_backend(oracle_backend):
    """BASELINE.json's nonlinear model with `fused=True` — the proposal net through `linear_gaussian.particle_mlp`
    (`_ops._ParticleMlp`: K13 forward, K13b backward; here the C oracle and float64 autograd behind the same provider
    interface) — against the same model with PyTorch modules: float64 loss and every gradient.  Holds the autograd wiring
    (which gradient goes to which of x, W1, the per-row offset, W2, b2; the shared-offset sum) without a GPU."""
    from aesmc_amd.testing.models import NonlinearSsm
    from aesmc_amd import _kernels
    provider = _kernels.get()
    calls = {"forward": 0, "backward": 0}
    forward, backward = provider.particle_mlp, provider.particle_mlp_backward

    def spy_forward(*args, **kwargs):
        calls["forward"] += 1
        return forward(*args, **kwargs)

    def spy_backward(*args, **kwargs):
        calls["backward"] += 1
        return backward(*args, **kwargs)
    results = []
    for fused in (False, True):
        model = NonlinearSsm(3, hidden=12, dtype=torch.float64, fused=fused)
        observations = model.simulate(4, 3, seed=2)
        torch.manual_seed(9)
        np.random.seed(9)
        provider.particle_mlp, provider.particle_mlp_backward = spy_forward, spy_backward
        try:
            loss = losses.get_loss(observations, 24, "aesmc", model.initial, model.transition, model.emission,
                                   model.proposal)
            loss.backward()
        finally:
            del provider.particle_mlp, provider.particle_mlp_backward
        results.append((loss.detach(), {name: p.grad.clone() for name, p in model.named_parameters()
                                        if p.grad is not None}))
    assert calls == {"forward": 3, "backward": 3}, calls
    (loss_a, grads_a), (loss_b, grads_b) = results
    assert abs(float(loss_a - loss_b)) <= 1e-12 * max(1.0, abs(float(loss_a)))
    assert sorted(grads_a) == sorted(grads_b)
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= 1e-9 * scale, name
