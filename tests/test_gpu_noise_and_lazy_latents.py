"""Noise formed inside kernels, propagation through the ancestor indices, lazily resampled latents (added in round 3):

  * the noise a kernel draws for itself IS `torch.empty(n).normal_()` — bit for bit, whatever the seed, the
    offset and the size, and the generator ends where PyTorch's own call would leave it;
  * the propagation launch that fetches x_{t-1} through the ancestor indices equals the resampling gather
    followed by the plain launch bit for bit (draw and log-weight), healthy and collapsed ancestries, ragged
    shapes, every piece width, out-of-range indices flagged and clamped;
  * a whole `infer` with the lazily resampled latent equals the run that gathers every step — every number.
"""
import numpy as np
import pytest
import torch

from tests.test_gpu_linear_gaussian import operands

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


# ---- noise -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,warm", [(0, 0), (1234, 3), (2 ** 40 + 17, 1)])
def test_philox_fill_is_torch_normal_bit_for_bit(kernels, hip_device, seed, warm):
    """aesmc_philox_normal_fill(seed, offset) against torch.empty(n).normal_() from the same generator state:
    every bit of every element, sizes below / at / across ATen's launch geometry (one thread per element, one
    trip, several trips with a ragged last one), and the offset PyTorch's call leaves behind predicted."""
    from aesmc_amd import _philox
    torch.cuda.init()
    generator = torch.cuda.default_generators[hip_device.index]
    torch.manual_seed(seed)
    for _ in range(warm):
        torch.randn(1000, device=hip_device)
    for numel in (1, 7, 255, 256, 1000, 2 ** 19 - 1, 2 ** 19 + 5, 5 * 2 ** 19 + 123, 256 * 1024 * 10,
                  1024 * 4096 * 10):
        offset = generator.get_offset()
        want = torch.empty(numel, device=hip_device).normal_()
        threads = _philox.launch_threads(numel, hip_device)
        assert generator.get_offset() == offset + _philox.consumed(numel, threads), numel
        got = torch.empty(numel, device=hip_device)
        status = kernels._lib.aesmc_philox_normal_fill(got.data_ptr(), numel, generator.initial_seed(), offset, threads,
                                                       0, None, torch.cuda.current_stream().cuda_stream)
        assert status == 0
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), numel


def test_reserving_noise_leaves_the_generator_where_normal_would(hip_device):
    """_philox.reserve(n) = the bookkeeping of torch.empty(n).normal_() without the launch: same descriptor,
    same offset afterwards, so everything drawn later is unchanged."""
    from aesmc_amd import _philox
    torch.manual_seed(77)
    torch.randn(10, device=hip_device)
    state = torch.cuda.get_rng_state(hip_device)
    reserved = _philox.reserve(3 * 4096 * 10, hip_device)
    after_reserve = torch.randn(5, device=hip_device)
    torch.cuda.set_rng_state(state, hip_device)
    drawn = torch.empty(3 * 4096 * 10, device=hip_device).normal_()
    after_draw = torch.randn(5, device=hip_device)
    assert torch.equal(after_reserve, after_draw)
    got = torch.empty_like(drawn)
    from aesmc_amd import _kernels
    status = _kernels.get()._lib.aesmc_philox_normal_fill(got.data_ptr(), got.numel(), reserved.seed, reserved.offset,
                                                          reserved.threads, 0, None, torch.cuda.current_stream().cuda_stream)
    assert status == 0 and torch.equal(got, drawn)


# ---- propagation through the ancestor indices --------------------------------------------------------
def _ancestors(B, K, device, seed, spread):
    """Sorted ancestor indices as systematic resampling makes them: `spread` = 1 healthy, large = collapsed."""
    from aesmc_amd import _ops
    gen = torch.Generator().manual_seed(seed)
    lw = (spread * torch.randn(B, K, generator=gen, dtype=torch.float64)).to(device)
    u = torch.rand(B, generator=gen, dtype=torch.float64).to(device)
    return _ops.ancestor_index(lw, u)


GATHER_SHAPES = [(3, 700, 10, 10), (2, 513, 5, 3), (5, 64, 16, 16), (1, 1000, 3, 7), (7, 300, 12, 2), (2, 2048, 8, 8),
                 (16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (40, 60, 6, 9), (3, 50, 1, 1), (2, 999, 9, 4)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("spread", [1.0, 6.0])
@pytest.mark.parametrize("shape", GATHER_SHAPES)
def test_propagate_through_ancestors_equals_gather_then_propagate(kernels, hip_device, dtype, spread, shape):
    """aesmc_affine_normal_propagate_resampled == aesmc_resample_gather, then aesmc_affine_normal_propagate: the
    draw and the log-weight bit for bit (rows of 4-, 8- and 16-byte pieces, one and two particles per lane,
    tiles that straddle batch rows, ragged tails, N(0,1) and heavily collapsed weights)."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=3 * B + K + dx)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=spread)
    off_p = torch.from_numpy(np.random.RandomState(6).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    moved = kernels.gather(o["x_prev"], idx)
    want_x = torch.full_like(moved, float("nan"))
    want_lw = kernels.affine_propagate(moved, o["eps"], o["y"], *terms, scales, out_x=want_x)
    got_x = torch.full_like(moved, float("nan"))
    got_lw = kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=got_x, ancestors=idx)
    assert got_lw is not None, "the gathering launch declined a shape it should cover"
    assert torch.equal(got_x, want_x)
    assert torch.equal(got_lw, want_lw)
    assert kernels.read_flags(hip_device) == 0


def test_propagate_through_ancestors_declines_tiny_rows_and_clamps_bad_indices(kernels, hip_device):
    from aesmc_amd import _lib
    B, K, dx, dy = 300, 5, 4, 4        # fewer than ~43 particles per batch row: the caller gathers first
    n, o = operands(B, K, dx, dy, np.float32, hip_device, seed=1)
    idx = _ancestors(B, K, hip_device, seed=2, spread=1.0)
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    out_x = torch.empty_like(o["x_prev"])
    assert kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=out_x, ancestors=idx) is None
    B, K, dx, dy = 4, 600, 10, 10
    n, o = operands(B, K, dx, dy, np.float32, hip_device, seed=4)
    idx = _ancestors(B, K, hip_device, seed=5, spread=1.0)
    bad = idx.clone()
    bad[1, 17] = K          # what K2 writes for a degenerate row
    bad[2, 0] = -3
    clamped = idx.clone()
    clamped[1, 17] = K - 1
    clamped[2, 0] = 0
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    want_x, got_x = torch.empty_like(o["x_prev"]), torch.empty_like(o["x_prev"])
    want = kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=want_x, ancestors=clamped)
    assert kernels.read_flags(hip_device) == 0
    got = kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=got_x, ancestors=bad)
    assert kernels.read_flags(hip_device) == _lib.FLAG_INDEX_OUT_OF_RANGE
    assert torch.equal(got, want) and torch.equal(got_x, want_x)


@pytest.mark.parametrize("grad", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_a_lazily_resampled_run_is_the_eagerly_gathered_run(hip_device, grad, dtype):
    """`infer` with the newest latent left un-gathered (the propagation launch fetches the rows) against the
    run whose resampling launch re-indexes it every step: latents, ancestors, evidence, gradients and both RNG
    streams identical; and the lazy run really launched no gather for the steps it fused."""
    from aesmc_amd import _kernels, inference
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    runs = {}
    for lazy in (False, True):
        inference.set_lazy_gather(lazy)
        calls = {"gather": 0, "propagate_ancestors": 0}
        real_gather, real_propagate = provider.gather, provider.affine_propagate

        def gather_spy(*args, **kwargs):
            calls["gather"] += 1
            return real_gather(*args, **kwargs)

        def propagate_spy(*args, **kwargs):
            calls["propagate_ancestors"] += kwargs.get("ancestors") is not None
            return real_propagate(*args, **kwargs)

        from aesmc_amd import state
        state.set_kernel_noise(False)       # this test is about the gather alone: torch draws the noise
        provider.gather, provider.affine_propagate = gather_spy, propagate_spy
        try:
            model = LgssmNd(10, dtype=dtype, affine=True).tune_proposal().to(hip_device)
            observations = model.simulate(6, 4, seed=3)
            torch.manual_seed(11)
            np.random.seed(11)
            with torch.set_grad_enabled(grad):
                out = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                      model.proposal, 1300, return_log_marginal_likelihood=True, return_latents=False,
                                      return_log_weight=not grad, return_ancestral_indices=True,
                                      return_original_latents=True)
            if grad:
                (-out["log_marginal_likelihood"].mean()).backward()
        finally:
            provider.gather, provider.affine_propagate = real_gather, real_propagate
            inference.set_lazy_gather(True)
            state.set_kernel_noise(True)
        after = (torch.rand(1, device=hip_device).item(), np.random.uniform())
        runs[lazy] = (out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, after, calls)
    (a, grads_a, rng_a, calls_a), (b, grads_b, rng_b, calls_b) = runs[False], runs[True]
    assert rng_a == rng_b
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    for x, y in zip(a["original_latents"] + a["ancestral_indices"], b["original_latents"] + b["ancestral_indices"]):
        assert torch.equal(x, y)
    assert torch.equal(a["last_latent"], b["last_latent"])
    assert sorted(grads_a) == sorted(grads_b) and (not grad or grads_a)
    for name in grads_a:
        assert torch.equal(grads_a[name], grads_b[name]), name
    assert calls_a["propagate_ancestors"] == 0 and calls_b["propagate_ancestors"] == 5
    if not grad:
        assert calls_b["gather"] == 0


def test_a_model_that_reads_the_resampled_latent_gets_the_gathered_values(hip_device):
    """Callables that need the VALUES of previous_latents[-1] (a tanh transition) materialise the lazy entry — the same
    numbers as the eager gather, bit for bit — and after the first such step `infer` goes back to gathering inside
    the resampling launch."""
    from aesmc_amd import inference
    from aesmc_amd._lazy import LazyResampled
    from aesmc_amd.testing.models import NonlinearSsm
    seen = []

    class Spy(NonlinearSsm):
        def transition(self, previous_latents=None, time=None, previous_observations=None):
            seen.append(type(previous_latents[-1]) is LazyResampled)
            return super().transition(previous_latents=previous_latents, time=time,
                                      previous_observations=previous_observations)

    outs = {}
    for lazy in (False, True):
        with inference.lazy_gather(lazy):
            model = Spy(5, hidden=16, dtype=torch.float64).to(hip_device)
            observations = model.simulate(5, 3, seed=2)
            torch.manual_seed(4)
            np.random.seed(4)
            del seen[:]
            outs[lazy] = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                         model.proposal, 300, return_log_marginal_likelihood=True,
                                         return_ancestral_indices=True)
            if lazy:
                assert seen == [True, False, False, False]
    assert torch.equal(outs[False]["log_marginal_likelihood"], outs[True]["log_marginal_likelihood"])
    for x, y in zip(outs[False]["latents"] + outs[False]["ancestral_indices"],
                    outs[True]["latents"] + outs[True]["ancestral_indices"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_a_reference_style_model_reaches_the_fused_kernels_unedited(hip_device, dtype):
    """VERDICT r02 item 5.  `LgssmNd(affine=False)` states its terms as the reference's own models do —
    `Normal(previous_latents[-1] @ W.t() + c, s)` (test/models/lgssm.py:40, :52, :66-77) — and is handed lazy latents:
    the affine expressions are RECORDED (`_lazy.LazyAffine`), `Normal(...)` built on them is recognised as a
    linear-Gaussian term, and every timestep after the first runs as ONE propagation launch forward (K16 / K15
    through the ancestors) and ONE launch backward (K14) — no matmul, no gather, no draw launch.  Against the same
    model evaluated by PyTorch itself: float64 ancestors identical, loss to 1e-12, gradients to 1e-9; float32 to
    rounding."""
    from aesmc_amd import _kernels, inference, losses
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    B, K, T, d = 4, 700, 6, 10
    counted = ("affine_propagate", "affine_propagate_drawn", "affine_step_backward", "gather", "affine_rsample",
               "particle_affine", "normal_logweight", "affine_initial_step")
    results = {}
    for lazy in (False, True):
        calls = dict.fromkeys(counted, 0)
        originals = {name: getattr(provider, name) for name in counted}
        for name in counted:
            def spy(*args, _name=name, **kwargs):
                out = originals[_name](*args, **kwargs)
                calls[_name] += out is not None
                return out
            setattr(provider, name, spy)
        try:
            model = LgssmNd(d, seed=0, dtype=dtype, affine=False).to(hip_device).tune_proposal()
            observations = [o.to(hip_device) for o in model.simulate(T, B, seed=1)]
            np.random.seed(2)
            torch.manual_seed(2)
            with inference.lazy_gather(lazy):
                loss = losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission,
                                       model.proposal)
            loss.backward()
        finally:
            for name, fn in originals.items():
                setattr(provider, name, fn)
        results[lazy] = (loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                         calls)
    (loss_a, grads_a, calls_a), (loss_b, grads_b, calls_b) = results[False], results[True]
    # PyTorch's evaluation: no fused launch at all;  the recorded one: one forward + one backward launch per step
    assert calls_a["affine_propagate"] == calls_a["affine_propagate_drawn"] == calls_a["affine_step_backward"] == 0
    assert calls_b["affine_propagate"] + calls_b["affine_propagate_drawn"] == T - 1
    assert calls_b["affine_step_backward"] == T - 1 and calls_b["gather"] == 0 and calls_b["affine_rsample"] == 0
    # time 0: one launch too (K20: the transposed draw, the emission's location, the log-weight) for float32 — the
    # emission's `latents[-1] @ C.t()` is recorded on the lazy first draw as well; float64: K6, K8 and K5
    assert (calls_b["affine_initial_step"], calls_b["normal_logweight"]) == ((1, 0) if dtype == torch.float32 else (0, 1))
    loss_tol, grad_tol = (1e-12, 1e-9) if dtype == torch.float64 else (2e-5, 5e-3)
    assert abs(float(loss_a - loss_b)) <= loss_tol * max(1.0, abs(float(loss_a)))
    assert sorted(grads_a) == sorted(grads_b)
    for name in grads_a:
        scale = max(float(grads_a[name].abs().max()), 1e-30)
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= grad_tol * scale, name


# ---- K16: the gather AND the noise inside the propagation launch -----------------------------------------------
DRAWN_SHAPES = [(3, 700, 10, 10), (16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (7, 300, 12, 2),
                (2, 2048, 8, 8), (2, 999, 9, 4), (5, 130, 16, 16), (2, 513, 5, 3), (1, 1000, 3, 7), (64, 16384, 2, 2),
                # sizes around ATen's launch geometry (G = 2^19 thread ids): one element per thread, a wrap of a
                # few elements into the next window, ragged last blocks, a second trip with one window only
                (1, 52428, 10, 10), (1, 52429, 10, 3), (5, 52430, 10, 10), (3, 174763, 3, 3), (1, 209716, 10, 10),
                (9, 65536, 4, 4),
                (1024, 4096, 10, 10)]      # the north-star shape itself


@pytest.mark.parametrize("gather", [True, False])
@pytest.mark.parametrize("shape", DRAWN_SHAPES)
def test_propagate_with_the_noise_inside_equals_normal_then_gather_then_propagate(kernels, hip_device, gather, shape,
                                                                                 monkeypatch):
    """aesmc_affine_normal_propagate_drawn == torch's own normal_(), aesmc_resample_gather, then
    aesmc_affine_normal_propagate — draw and log-weight bit for bit, and the generator where normal_ leaves it."""
    from aesmc_amd import _philox
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)      # the size policy is not what is tested here
    B, K, dx, dy = shape
    n, o = operands(min(B, 8), min(K, 64), dx, dy, np.float32, hip_device, seed=3 * B + K + dx)
    gen = torch.Generator(device=hip_device).manual_seed(B + K)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=1.0) if gather else None
    off_p = torch.from_numpy(np.random.RandomState(6).randn(dx).astype(np.float32)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], off_q))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    torch.manual_seed(1000 + K)
    torch.randn(17, device=hip_device)                      # some offset into the stream
    state = torch.cuda.get_rng_state(hip_device)
    eps = torch.empty(B, K, dx, device=hip_device).normal_()
    after_normal = torch.cuda.default_generators[hip_device.index].get_offset()
    torch.cuda.set_rng_state(state, hip_device)
    reservation = _philox.reserve(B * K * dx, hip_device)
    assert torch.cuda.default_generators[hip_device.index].get_offset() == after_normal
    moved = kernels.gather(x_prev, idx) if gather else x_prev
    want_x = torch.full_like(moved, float("nan"))
    want_lw = kernels.affine_propagate(moved, eps, y, *terms, scales, out_x=want_x)
    got_x = torch.full_like(moved, float("nan"))
    got_lw = kernels.affine_propagate_drawn(x_prev, reservation, y, *terms, scales, out_x=got_x, ancestors=idx)
    assert got_lw is not None, "the launch declined a shape it should cover"
    assert torch.equal(got_x, want_x)
    assert torch.equal(got_lw, want_lw)
    assert kernels.read_flags(hip_device) == 0
    assert torch.equal(kernels.philox_normal(reservation, (B, K, dx), hip_device), eps)


def test_propagate_with_the_noise_inside_declines_what_it_does_not_cover(kernels, hip_device, monkeypatch):
    from aesmc_amd import _philox
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)
    for B, K, dx, dy in ((300, 50, 4, 4), (3, 700, 1, 1)):      # short batch rows; one value per particle
        n, o = operands(B, K, dx, dy, np.float32, hip_device, seed=1)
        terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
        scales = (o["s_p"], o["s_g"], o["s_q"])
        reservation = _philox.reserve(B * K * dx, hip_device)
        out_x = torch.empty_like(o["x_prev"])
        assert kernels.affine_propagate_drawn(o["x_prev"], reservation, o["y"], *terms, scales, out_x=out_x) is None
    n, o = operands(4, 600, 10, 10, np.float64, hip_device, seed=1)       # float64: PyTorch draws it by another route
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    reservation = _philox.reserve(4 * 600 * 10, hip_device)
    assert kernels.affine_propagate_drawn(o["x_prev"], reservation, o["y"], *terms, (o["s_p"], o["s_g"], o["s_q"]),
                                          out_x=torch.empty_like(o["x_prev"])) is None


@pytest.mark.parametrize("grad", [False, True])
@pytest.mark.parametrize("learn_scales", [False, True])
@pytest.mark.parametrize("policy", ["inside the propagation launch", "filled by this library's own launch"])
def test_a_run_whose_kernels_draw_the_noise_is_the_run_that_lets_torch_draw_it(hip_device, grad, learn_scales, policy):
    """`infer` with the deferred draws' noise formed inside K16 from PyTorch's Philox stream against the run in
    which `_standard_normal` materialises it: latents, ancestors, evidence, gradients (also of learned scales)
    identical bit for bit, and both random streams end in the same state."""
    from aesmc_amd import _kernels, inference, state
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    threshold_was = type(provider).DRAWN_MIN_PARTICLES
    # (the size policy is a class constant: 0 since round 5 — the launch forms the noise at every size —; the other
    #  route, a fill launch of this library in front of the launch that reads the noise, is pinned by a high threshold)
    type(provider).DRAWN_MIN_PARTICLES = 0 if policy.startswith("inside") else 1 << 40
    runs = {}
    for inside in (False, True):
        state.set_kernel_noise(inside)
        calls = {"drawn": 0, "filled": 0}
        real, real_fill = provider.affine_propagate_drawn, provider.philox_normal

        def spy(*args, **kwargs):
            out = real(*args, **kwargs)
            calls["drawn"] += out is not None
            return out

        def fill_spy(*args, **kwargs):
            calls["filled"] += 1
            return real_fill(*args, **kwargs)

        provider.affine_propagate_drawn, provider.philox_normal = spy, fill_spy
        try:
            model = LgssmNd(10, dtype=torch.float32, affine=True).tune_proposal().to(hip_device)
            if learn_scales:
                for name in ("transition_scale", "emission_scale", "proposal_scale"):
                    value = getattr(model, name).detach().clone()
                    delattr(model, name)
                    model.register_parameter(name, torch.nn.Parameter(value))
            observations = model.simulate(6, 4, seed=3)
            torch.manual_seed(11)
            np.random.seed(11)
            with torch.set_grad_enabled(grad):
                out = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                      model.proposal, 1300, return_log_marginal_likelihood=True, return_latents=False,
                                      return_log_weight=not grad, return_ancestral_indices=True,
                                      return_original_latents=True)
            if grad:
                (-out["log_marginal_likelihood"].mean()).backward()
        finally:
            provider.affine_propagate_drawn, provider.philox_normal = real, real_fill
            state.set_kernel_noise(True)
            if inside:
                type(provider).DRAWN_MIN_PARTICLES = threshold_was
        after = (torch.rand(1, device=hip_device).item(), np.random.uniform())
        runs[inside] = (out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, after, calls)
    (a, grads_a, rng_a, calls_a), (b, grads_b, rng_b, calls_b) = runs[False], runs[True]
    assert calls_a == {"drawn": 0, "filled": 0}
    assert calls_b == ({"drawn": 5, "filled": 0} if policy.startswith("inside") else {"drawn": 0, "filled": 5})
    assert rng_a == rng_b
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    for x, y in zip(a["original_latents"] + a["ancestral_indices"], b["original_latents"] + b["ancestral_indices"]):
        assert torch.equal(x, y)
    assert sorted(grads_a) == sorted(grads_b) and (not grad or grads_a)
    if grad and learn_scales:
        assert "proposal_scale" in grads_a
    for name in grads_a:
        assert torch.equal(grads_a[name], grads_b[name]), name


def test_replayed_noise_bypasses_the_kernel_noise(hip_device):
    """A test harness that replays recorded normals through torch.distributions.normal._standard_normal must see
    them used: with that function replaced, nothing is reserved and the noise arrives as the tensor it returns."""
    from aesmc_amd import _kernels, inference
    from aesmc_amd.testing import replay
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    model = LgssmNd(10, dtype=torch.float32, affine=True).tune_proposal().to(hip_device)
    observations = model.simulate(4, 3, seed=3)
    torch.manual_seed(5)
    np.random.seed(5)
    with replay.record() as tape, torch.no_grad():
        first = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal,
                                700, return_log_marginal_likelihood=True)
    assert len(tape.normals) == 4
    calls = {"drawn": 0}
    real = provider.affine_propagate_drawn
    provider.affine_propagate_drawn = lambda *a, **k: calls.__setitem__("drawn", calls["drawn"] + 1) or real(*a, **k)
    try:
        with replay.replay(tape), torch.no_grad():
            again = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                    model.proposal, 700, return_log_marginal_likelihood=True)
    finally:
        provider.affine_propagate_drawn = real
    assert calls["drawn"] == 0
    assert torch.equal(first["log_marginal_likelihood"], again["log_marginal_likelihood"])


# ---- K14 through the ancestor indices ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(3, 700, 10, 10), (2, 513, 5, 3), (5, 64, 16, 16), (7, 300, 12, 2), (2, 2048, 8, 8),
                                   (16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (2, 999, 9, 4)])
def test_step_backward_through_ancestors_equals_gather_then_step_backward(kernels, hip_device, dtype, shape):
    """aesmc_affine_step_backward_resampled(x_src, ancestors) == aesmc_resample_gather, then
    aesmc_affine_step_backward: every gradient bit for bit (the gradient of the resampled rows included)."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=3 * B + K + dx)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=2.0)
    off_p = torch.from_numpy(np.random.RandomState(4).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    moved = kernels.gather(o["x_prev"], idx)
    x = kernels.affine_rsample(moved, o["Q"], o["off_q"], o["eps"], o["s_q"])
    lw = kernels.affine_logweight(moved, x, o["y"], *terms, scales)
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    rng = np.random.RandomState(9)
    grad_lse = torch.from_numpy(rng.randn(B).astype(dtype)).to(hip_device)
    grad_x = torch.from_numpy(rng.randn(B, K, dx).astype(dtype)).to(hip_device)
    need = [True] * 12
    need[1] = False
    want = kernels.affine_step_backward(moved, x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse, grad_x=grad_x)
    got = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                       grad_x=grad_x, ancestors=idx)
    for a, b in zip(got, want):
        assert (a is None and b is None) or torch.equal(a, b)
    assert kernels.read_flags(hip_device) == 0


# ---- torch.gather's backward inside K14 ----------------------------------------------------
def _next_resampling(kernels, B, K, device, seed, spread, dtype=torch.float64):
    """The NEXT step's resampling launch with the children ranges as its by-product: (indices, child_end)."""
    gen = torch.Generator().manual_seed(seed)
    lw = (spread * torch.randn(B, K, generator=gen, dtype=torch.float64)).to(dtype).to(device)
    u = torch.rand(B, generator=gen, dtype=torch.float64).to(device)
    plain = kernels.resample_step(lw, u, None, want_lse=True)
    ranged = kernels.resample_step(lw, u, None, want_lse=True, want_child_end=True)
    assert torch.equal(plain[0], ranged[0]) and torch.equal(plain[1], ranged[1])
    return ranged[0], ranged[0]._aesmc_child_end


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("spread", [0.0, 1.0, 9.0])
@pytest.mark.parametrize("shape", [(3, 700), (2, 513), (5, 64), (300, 4096), (16, 10000), (2, 30000), (7, 2), (4, 1)])
def test_the_resampling_launch_says_where_each_particles_children_end(kernels, hip_device, dtype, spread, shape):
    """aesmc_resample_step_ranges: the indices and the row log-sum-exp of aesmc_resample_step, bit for bit, and
    child_end[b,k] = #{k' : idx[b,k'] <= k} — uniform, N(0,1) and collapsed weights, every particles-per-lane
    instantiation, K = 1."""
    B, K = shape
    idx, child_end = _next_resampling(kernels, B, K, hip_device, seed=B + K, spread=spread, dtype=dtype)
    assert child_end.dtype == torch.int32 and child_end.shape == (B, K)
    host = idx.cpu().numpy()
    want = np.stack([np.searchsorted(host[b], np.arange(K), side="right") for b in range(B)])
    assert np.array_equal(child_end.cpu().numpy(), want)
    assert int(child_end[:, -1].min()) == K == int(child_end[:, -1].max())
    assert kernels.read_flags(hip_device) == 0


def test_a_degenerate_row_has_no_children(kernels, hip_device):
    """A row of -inf weights: the launch flags it, its indices are K (as without the ranges) and nobody has children."""
    from aesmc_amd import _lib
    B, K = 3, 600
    lw = torch.randn(B, K, dtype=torch.float64, device=hip_device)
    lw[1] = float("-inf")
    u = torch.full((B,), 0.3, dtype=torch.float64, device=hip_device)
    idx, _, _ = kernels.resample_step(lw, u, None, want_lse=False, want_child_end=True)
    assert kernels.read_flags(hip_device) & _lib.FLAG_DEGENERATE_ROW
    child_end = idx._aesmc_child_end
    assert int(child_end[1].abs().max()) == 0 and bool((idx[1] == K).all())
    assert int(child_end[0, -1]) == K and int(child_end[2, -1]) == K


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("spread", [1.0, 9.0])
@pytest.mark.parametrize("with_grad_x", [False, True])
@pytest.mark.parametrize("shape", [(3, 700, 10, 10), (2, 513, 5, 3), (5, 64, 16, 16), (7, 300, 12, 2), (2, 2048, 8, 8),
                                   (16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (2, 999, 9, 4)])
def test_step_backward_sums_the_children_of_its_particles_itself(kernels, hip_device, dtype, spread, with_grad_x, shape):
    """aesmc_affine_step_backward_resampled(child_grad, child_end) == aesmc_resample_gather_backward of the next
    step's per-child gradient, added to whatever else reaches x_t, then the same entry point with the sum as
    `grad_x`: every gradient to rounding (a particle's children are added in k order here, in the segmented-sum
    kernel's order there), healthy and collapsed next-step ancestries (runs longer than a lane takes alone), and
    against the sum taken in float64."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=3 * B + K + dx)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=2.0)
    off_p = torch.from_numpy(np.random.RandomState(4).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    moved = kernels.gather(o["x_prev"], idx)
    x = kernels.affine_rsample(moved, o["Q"], o["off_q"], o["eps"], o["s_q"])
    lw = kernels.affine_logweight(moved, x, o["y"], *terms, scales)
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    rng = np.random.RandomState(9)
    grad_lse = torch.from_numpy(rng.randn(B).astype(dtype)).to(hip_device)
    grad_x = torch.from_numpy(rng.randn(B, K, dx).astype(dtype)).to(hip_device) if with_grad_x else None
    child_grad = torch.from_numpy(rng.randn(B, K, dx).astype(dtype)).to(hip_device)
    idx_next, child_end = _next_resampling(kernels, B, K, hip_device, seed=7 * B + K, spread=spread)
    summed = kernels.gather_backward(child_grad, idx_next, sorted_index=True)
    flat = (idx_next + K * torch.arange(B, device=hip_device).unsqueeze(1)).reshape(-1)
    exact = torch.zeros(B * K, dx, dtype=torch.float64, device=hip_device).index_add_(
        0, flat, child_grad.double().reshape(B * K, dx)).view(B, K, dx)
    need = [True] * 12
    need[1] = False
    arrives = summed if grad_x is None else grad_x + summed
    want = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                        grad_x=arrives, ancestors=idx)
    arrives64 = (exact if grad_x is None else grad_x.double() + exact).to(arrives.dtype)
    want64 = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                          grad_x=arrives64, ancestors=idx)
    got = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                       grad_x=grad_x, ancestors=idx, child_grad=child_grad, child_end=child_end)
    again = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                         grad_x=grad_x, ancestors=idx, child_grad=child_grad, child_end=child_end)
    tolerance = 3e-5 if dtype == np.float32 else 1e-11
    names = ("x_prev", "x", "y", "A", "off_p", "C", "off_g", "Q", "off_q", "s_p", "s_g", "s_q")
    longest = int((child_end[:, 1:] - child_end[:, :-1]).max()) if K > 1 else K
    assert spread < 5 or longest > 32, "the collapsed case is meant to reach runs the wavefront shares out"
    for name, a, b, c, d in zip(names, got, want, want64, again):
        if name == "x":
            assert a is None and b is None
            continue
        assert torch.equal(a, d), (name, "not reproducible")
        scale = max(1.0, float(c.abs().max())) * (1.0 + np.sqrt(longest))
        assert float((a.double() - b.double()).abs().max()) <= tolerance * scale, (name, "vs the launch it replaces")
        assert float((a.double() - c.double()).abs().max()) <= tolerance * scale, (name, "vs the float64 sum")
    assert kernels.read_flags(hip_device) == 0
    # only the children's gradient arrives (no ELBO term of this step, nothing else at x_t)
    only = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, ancestors=idx,
                                        child_grad=child_grad, child_end=child_end)
    want_only = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_x=summed,
                                             ancestors=idx)
    for name, a, b in zip(names, only, want_only):
        if name != "x":
            scale = max(1.0, float(b.abs().max())) * (1.0 + np.sqrt(longest))
            assert float((a.double() - b.double()).abs().max()) <= tolerance * scale, (name, "children only")


def test_step_backward_declines_children_without_ancestors(kernels, hip_device):
    n, o = operands(2, 300, 4, 3, np.float32, hip_device, seed=1)
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    lw = kernels.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    need = [True] * 12
    need[1] = False
    _, child_end = _next_resampling(kernels, 2, 300, hip_device, seed=3, spread=1.0)
    with pytest.raises(ValueError):
        kernels.affine_step_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, lw, lse,
                                     child_grad=torch.ones_like(o["x"]), child_end=child_end)
    idx = _ancestors(2, 300, hip_device, seed=5, spread=1.0)
    with pytest.raises(ValueError):      # ranges of another shape / dtype
        kernels.affine_step_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, lw, lse, ancestors=idx,
                                     child_grad=torch.ones_like(o["x"]), child_end=child_end.long())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("learn_scales", [False, True])
def test_a_loss_whose_steps_hand_the_gather_backward_on_is_the_loss_that_launches_it(hip_device, dtype, learn_scales):
    """get_loss + backward with consecutive steps' nodes linked (`_ops.StepLink`: K14 sums each particle's children
    while it consumes them) against the run whose every step launches the segmented sum: the loss bit for bit,
    every parameter gradient to rounding, and the linked run launches the gather's backward once (for x_0, whose
    producer is not a step node) instead of T - 1 times.  A second backward through the retained graph gives the
    same gradients (the deposits are taken, not kept)."""
    from aesmc_amd import _kernels, inference, losses
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    T, B, K = 7, 3, 1500
    runs = {}
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
            model = LgssmNd(10, dtype=dtype, affine=True).tune_proposal().to(hip_device)
            if learn_scales:
                for name in ("transition_scale", "emission_scale", "proposal_scale"):
                    value = getattr(model, name).detach().clone()
                    delattr(model, name)
                    model.register_parameter(name, torch.nn.Parameter(value))
            observations = model.simulate(T, B, seed=3)
            torch.manual_seed(11)
            np.random.seed(11)
            with inference.fold_gather_backward(fold):
                loss = losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission,
                                       model.proposal)
                loss.backward(retain_graph=True)
                first = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
                counted = dict(calls)
                for p in model.parameters():
                    p.grad = None
                loss.backward()
                second = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            provider.gather_backward, provider.affine_step_backward = real_gb, real_sb
        runs[fold] = (loss.detach(), first, second, counted)
    (loss_a, grads_a, again_a, calls_a), (loss_b, grads_b, again_b, calls_b) = runs[False], runs[True]
    assert torch.equal(loss_a, loss_b)
    assert calls_a == {"gather_backward": T - 1, "with_children": 0}
    assert calls_b == {"gather_backward": 1, "with_children": T - 2}
    assert sorted(grads_a) == sorted(grads_b) and grads_a
    tolerance = 2e-4 if dtype == torch.float32 else 1e-10
    for name in grads_a:
        scale = max(1.0, float(grads_a[name].abs().max()))
        assert float((grads_a[name] - grads_b[name]).abs().max()) <= tolerance * scale, name
        assert torch.equal(grads_b[name], again_b[name]), (name, "second backward")
        assert torch.equal(grads_a[name], again_a[name]), (name, "second backward, unlinked")


def test_latents_handed_to_the_caller_keep_every_step_its_own_gather_backward(hip_device):
    """`infer(return_original_latents=True)`: somebody outside may differentiate x_t itself, so no step hands its
    gather's backward to its predecessor — x_t's gradient is the full one (checked against the unlinked run)."""
    from aesmc_amd import _kernels, inference
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    grads = {}
    for fold in (False, True):
        calls = {"with_children": 0}
        real_sb = provider.affine_step_backward
        provider.affine_step_backward = lambda *a, **k: (calls.__setitem__(
            "with_children", calls["with_children"] + (k.get("child_grad") is not None)), real_sb(*a, **k))[1]
        try:
            model = LgssmNd(10, dtype=torch.float64, affine=True).tune_proposal().to(hip_device)
            observations = model.simulate(5, 2, seed=3)
            torch.manual_seed(2)
            np.random.seed(2)
            with inference.fold_gather_backward(fold):
                out = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                      model.proposal, 900, return_log_marginal_likelihood=True, return_latents=False,
                                      return_log_weight=False, return_original_latents=True)
            middle = out["original_latents"][2]
            grads[fold] = torch.autograd.grad(out["log_marginal_likelihood"].sum(), middle)[0]
        finally:
            provider.affine_step_backward = real_sb
        assert calls["with_children"] == 0
    assert torch.equal(grads[False], grads[True])


# ---- the headline route against the CPU port at the north-star batch --------------------------------------------------
def test_the_north_star_shape_matches_the_cpu_port_draw_for_draw(hip_device):
    """B=1024, K=4096, d=10 — the north-star shape — three timesteps, float64: the CPU port of the reference (plain `Normal(matmul)` callables,
    PyTorch autograd on the host) records its draws; the product replays them through the route `get_loss` takes at
    the bench shape — K2 with the children ranges, propagation through the ancestors over persistent workgroups that
    walk several tiles each, K14 with the gather's backward folded in.  Every ancestor index equal, loss to 1e-10,
    every parameter gradient to 1e-8 of its largest entry.  (The float32 launch that also draws the noise is held to
    this route bit for bit at the north-star shape by test_propagate_with_the_noise_inside_...[shape17].)"""
    from aesmc_amd import _kernels, inference, losses
    from aesmc_amd.testing import models, replay
    from oracle import reference_port
    B, K, T, d = 1024, 4096, 3, 10
    dtype = torch.float64
    cpu_model = models.LgssmNd(d, seed=0, dtype=dtype, state=reference_port).tune_proposal()
    observations = cpu_model.simulate(T, B, seed=1)
    parts = lambda m: (m.initial, m.transition, m.emission, m.proposal)
    np.random.seed(7)
    torch.manual_seed(7)
    want_indices, real_sampler = [], reference_port.sample_ancestral_index

    def sampler_spy(log_weight):
        want_indices.append(real_sampler(log_weight))
        return want_indices[-1]

    reference_port.sample_ancestral_index = sampler_spy
    try:
        with replay.record() as tape:
            want = reference_port.get_loss(observations, K, "aesmc", *parts(cpu_model))
    finally:
        reference_port.sample_ancestral_index = real_sampler
    want.backward()
    model = models.LgssmNd(d, seed=0, dtype=dtype, affine=True).to(hip_device).tune_proposal()
    device_observations = [o.to(hip_device) for o in observations]
    provider = _kernels.get()
    calls = {"with_children": 0, "through_ancestors": 0}
    got_indices = []
    real_sb, real_prop, real_step = provider.affine_step_backward, provider.affine_propagate, provider.resample_step

    def step_spy(*args, **kwargs):
        out = real_step(*args, **kwargs)
        got_indices.append(out[0])
        return out

    def sb_spy(*args, **kwargs):
        calls["with_children"] += kwargs.get("child_grad") is not None
        return real_sb(*args, **kwargs)

    def prop_spy(*args, **kwargs):
        calls["through_ancestors"] += kwargs.get("ancestors") is not None
        return real_prop(*args, **kwargs)

    provider.affine_step_backward, provider.affine_propagate, provider.resample_step = sb_spy, prop_spy, step_spy
    try:
        with replay.replay(tape):
            got = losses.get_loss(device_observations, K, "aesmc", *parts(model))
        got.backward()
    finally:
        provider.affine_step_backward, provider.affine_propagate, provider.resample_step = real_sb, real_prop, real_step
    assert calls["with_children"] == T - 2 and calls["through_ancestors"] == T - 1
    assert len(got_indices) == len(want_indices) == T - 1
    for a, b in zip(got_indices, want_indices):
        assert torch.equal(a.cpu(), b)
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-10, atol=1e-10)
    expected = dict(cpu_model.named_parameters())
    for name, parameter in model.named_parameters():
        reference = expected[name].grad
        if reference is None:
            assert parameter.grad is None or float(parameter.grad.abs().max()) == 0.0, name
            continue
        assert parameter.grad is not None, name
        scale = max(float(reference.abs().max()), 1e-30)
        assert float((parameter.grad.cpu() - reference).abs().max()) <= 1e-8 * scale, name


# ---- float32 runs of the reference itself, end to end --------------------------------------
# What the device achieved on the committed fixtures when these bounds were written (MI355X, both model statements):
# see profiles/README.md "float32 fixture parity"; the bounds below are those numbers plus a margin.
FP32_BOUNDS = {
    # name: (teacher-forced max rel |d log w| over ALL steps, teacher-forced flip rate,
    #        free-running mean index agreement (None: the first flip comes in the first step and the two particle
    #        systems are different ones from there on — index agreement then measures nothing), free-running rel |d log Z|)
    # achieved on MI355X when written (profiles/README.md "float32 fixture parity"): forced d log w 5.0e-7 .. 9.2e-7;
    # flips 2 of 38 912 (K=1024), 73-75 of 73 728 (K=4096), 0 elsewhere; free-running rel d log Z 0 (K=1024),
    # 4.3e-3 .. 4.5e-3 (K=4096), <= 1.3e-7 elsewhere
    # (K=1024: the free-running numbers depend on which GEMM computes the callables' small matmuls — under bench.py's
    #  TunableOp picks 4 indices flip instead of 2 and log Z moves by 1.6e-3 — so they are bounded like K=4096's)
    "lgssm10d_k1024_smc_f32": (5e-6, 3e-4, None, 1e-2),
    "lgssm10d_k4096_smc_f32": (5e-6, 2e-3, None, 1e-2),
    "lgssm3d_smc_f32": (5e-6, 4e-4, 0.9995, 1e-5),
    "c1_lgssm1d_smc_f32": (5e-6, 0.0, 1.0, 1e-6),
    "c1_lgssm1d_smc_stock_f32": (5e-6, 0.0, 1.0, 1e-6),
}


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("name", sorted(FP32_BOUNDS))
def test_float32_runs_of_the_reference_teacher_forced_and_free_running(hip_device, name, affine, capsys):
    """Every float32 SMC run captured from the reference, replayed on the device draw for draw.  Teacher-forced (the
    reference's ancestors substituted after each resampling launch) EVERY step's log-weights agree to float32
    rounding — also after the first flip — and the launch's own indices differ from the reference's float32-CDF
    indices at a rate under SURVEY section 7's; free-running, agreement and log Z stay within the measured bounds.
    The achieved numbers are printed (pytest -s) and go into bench.py's JSON."""
    from aesmc_amd import state
    from aesmc_amd.testing import parity
    from tests.golden_io import Golden, float32_flip_bound
    case = Golden(name)
    if affine and case.meta["model"] != "lgssm_nd":
        pytest.skip("only the d-dimensional LGSSM has an AffineNormal statement")
    parts, _ = case.build_parts(state, hip_device, affine=affine)
    steps = case.meta["num_timesteps"] - 1
    reference = {"log_weights": case.series("out_log_weights"), "indices": case.series("out_idx")[:steps],
                 "lml": case["out_lml"]}
    got = parity.float32_fixture_parity(parts, case.observations(hip_device), case.meta["num_particles"], case.tape(),
                                        reference)
    with capsys.disabled():
        print("\\n[fp32 parity] {} affine={}: forced max rel dlogw {:.2e}, flips {} of {} ({:.2e}), forced rel dlogZ {:.2e}; "
              "free agreement min {:.4f} mean {:.4f}, first flip at step {}, rel dlogZ {:.2e}".format(
                  name, affine, got["teacher_forced_max_rel_dlogw"], sum(got["teacher_forced_flips_per_step"]),
                  steps * reference["indices"][0].size, got["teacher_forced_flip_rate"], got["teacher_forced_rel_dlogZ"],
                  got["free_running_index_agreement_min"], got["free_running_index_agreement_mean"],
                  got["free_running_first_flip_step"], got["free_running_rel_dlogZ"]))
    lw_bound, flip_bound, agreement_bound, lml_bound = FP32_BOUNDS[name]
    assert got["teacher_forced_max_rel_dlogw"] <= lw_bound
    assert got["teacher_forced_flip_rate"] <= max(flip_bound, 0.0)
    assert got["teacher_forced_flip_rate"] <= float32_flip_bound(case.meta["num_particles"])
    assert got["teacher_forced_rel_dlogZ"] <= 2e-6
    if agreement_bound is not None:
        assert got["free_running_index_agreement_mean"] >= agreement_bound
    assert got["free_running_rel_dlogZ"] <= lml_bound


def test_uniforms_read_from_the_pinned_block_are_the_uniforms_copied_to_the_device(hip_device):
    """The resampling launch reading its per-row uniforms out of the pinned block the host wrote them into
    (`aesmc_host_device_pointer`, no copy launch per timestep) against the run that copies each block to the device:
    the same ancestors, evidence and latents; and the mapped view really is the host block."""
    from aesmc_amd import inference
    from aesmc_amd.testing.models import LgssmNd
    feed = inference._UniformFeed(16, 3, hip_device)
    assert feed.mapped_rows is not None, "the device does not map PyTorch's pinned memory: the copy path is in use"
    np.random.seed(5)
    row = feed.next()
    assert row.is_cuda and np.array_equal(row.cpu().numpy(), feed.host_np[0])
    runs = {}
    for mapped in (False, True):
        was = inference._ZERO_COPY_UNIFORMS
        inference._ZERO_COPY_UNIFORMS = mapped
        try:
            model = LgssmNd(10, dtype=torch.float32, affine=True).tune_proposal().to(hip_device)
            observations = model.simulate(12, 5, seed=3)
            torch.manual_seed(11)
            np.random.seed(11)
            with torch.no_grad():
                runs[mapped] = inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                               model.proposal, 900, return_log_marginal_likelihood=True,
                                               return_ancestral_indices=True)
        finally:
            inference._ZERO_COPY_UNIFORMS = was
    a, b = runs[False], runs[True]
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    for x, y in zip(a["ancestral_indices"] + a["latents"], b["ancestral_indices"] + b["latents"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(3, 700, 10, 10), (2, 513, 5, 3), (5, 64, 16, 16), (300, 4096, 10, 10), (520, 2100, 8, 8),
                                   (1024, 4096, 10, 10)])
def test_steps_that_share_their_parameters_finish_their_gradients_once(kernels, hip_device, dtype, shape):
    """aesmc_affine_chain: a deferring call writes no weight / scale gradient and leaves records; collected on their
    own (aesmc_affine_backward_collect) they are bit for bit what the call gives when it finishes them itself; a
    second call that carries them returns the two steps' sum (to rounding: the steps are added record by record
    before the workgroups are, not after), reproducibly; the per-step outputs (the gradient of x_{t-1}'s rows, the
    offsets' and y's) of a chained call are bit for bit the unchained call's."""
    B, K, dx, dy = shape
    if dtype == np.float64 and B * K > (1 << 21):
        pytest.skip("the float32 case covers the full size")
    names = ("x_prev", "x", "y", "A", "off_p", "C", "off_g", "Q", "off_q", "s_p", "s_g", "s_q")
    shared = ("A", "C", "Q", "s_p", "s_g", "s_q")
    need = [True] * 12
    need[1] = False
    steps = []
    for step in range(2):
        n, o = operands(B, K, dx, dy, dtype, hip_device, seed=3 * B + K + dx + 100 * step)
        if step == 1:       # the same parameters as the step before, everything else its own
            for name in shared + ("off_g",):
                o[name] = steps[0]["o"][name]
        idx = _ancestors(B, K, hip_device, seed=B + K + step, spread=2.0)
        terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
        scales = (o["s_p"], o["s_g"], o["s_q"])
        moved = kernels.gather(o["x_prev"], idx)
        x = kernels.affine_rsample(moved, o["Q"], o["off_q"], o["eps"], o["s_q"])
        lw = kernels.affine_logweight(moved, x, o["y"], *terms, scales)
        _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
        grad_lse = torch.from_numpy(np.random.RandomState(9 + step).randn(B).astype(dtype)).to(hip_device)
        call = lambda chain, o=o, x=x, terms=terms, scales=scales, lw=lw, lse=lse, grad_lse=grad_lse, idx=idx: \
            kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                         ancestors=idx, chain=chain)
        steps.append({"o": o, "call": call, "alone": call(None)})
    later, earlier = steps[1], steps[0]      # backward order: the later step runs first and defers
    chain = {"carry": None, "defer": True}
    deferred = later["call"](chain)
    left = chain["left"]
    if left is None:
        pytest.skip("K14 declines the shape: nothing is deferred (the unfused route returns every gradient)")
    assert left[1] >= 1
    for name, got, alone in zip(names, deferred, later["alone"]):
        if name in shared:
            assert got is None, name
        elif alone is not None:
            assert torch.equal(got, alone), name
    collected = kernels.affine_backward_collect(left, deferred[0].dtype, hip_device, dx, dy, need,
                                                (later["o"]["s_p"], later["o"]["s_g"], later["o"]["s_q"]))
    for name, got, alone in zip(names, collected, later["alone"]):
        if name in shared:
            assert torch.equal(got, alone), (name, "collected")
    results = []
    for _ in range(2):
        chain = {"carry": left, "defer": False}
        results.append(earlier["call"](chain))
        assert chain["left"] is None
    tolerance = 2e-5 if dtype == np.float32 else 1e-12
    for name, got, again, a, b in zip(names, results[0], results[1], earlier["alone"], later["alone"]):
        if name == "x" or a is None:
            continue
        assert torch.equal(got, again), (name, "not reproducible")
        if name in shared:
            want = a.double() + b.double()
            # (sums of B K terms of either sign: the bound is on the terms' size, which the sum of the two results' does not show)
            scale = max(1.0, float(want.abs().max()), float(a.abs().max()), float(b.abs().max())) * np.sqrt(B * K / 4096.0 + 1.0)
            assert float((got.double() - want).abs().max()) <= tolerance * scale, name
        else:
            assert torch.equal(got, a), name
    # carrying AND deferring (a step in the middle of a run): records again, which collect to the same sums
    chain = {"carry": left, "defer": True}
    middle = earlier["call"](chain)
    assert all(middle[names.index(name)] is None for name in shared)
    # the run's interleaved weight pairs are built by its first call and handed on with the records (rows form only)
    if len(left) > 2 and left[2]:
        assert chain["left"][2] == left[2] and chain["left"][3] is left[3], "the pairs were rebuilt instead of handed on"
    both = kernels.affine_backward_collect(chain["left"], deferred[0].dtype, hip_device, dx, dy, need,
                                           (earlier["o"]["s_p"], earlier["o"]["s_g"], earlier["o"]["s_q"]))
    for name, got, want in zip(names, both, results[0]):
        if name in shared:
            assert torch.equal(got, want), (name, "carried and deferred")
    assert kernels.read_flags(hip_device) == 0


@pytest.mark.parametrize("shape", [(64, 4096, 10), (3, 70001, 5), (128, 4096, 1), (7, 1999, 33), (2, 40000, 128),
                                   (1024, 4096, 10), (5, 300, 7)])
@pytest.mark.parametrize("layout", ["dense", "row_loc", "column_scale", "dense_scale"])
def test_a_plain_normals_draw_forms_its_noise_in_the_launch(kernels, hip_device, shape, layout):
    """`state.sample` of a FULLY_EXPANDED Normal with tensor parameters: above a quarter of a million elements the
    noise is formed inside the launch that adds the location (aesmc_normal_rsample_drawn) — the draw equals, bit for
    bit, what `torch.empty(shape).normal_()` followed by K6 gives under the same seed, the generator ends where
    `normal_` would have left it, and the gradient reaches the location."""
    from aesmc_amd import _philox, state
    B, K, D = shape
    gen = torch.Generator().manual_seed(B + K + D)
    loc = torch.randn(B, K, D, generator=gen).to(hip_device)
    scale = torch.tensor(0.7, device=hip_device)
    if layout == "row_loc":
        loc = torch.randn(B, 1, D, generator=gen).to(hip_device).expand(B, K, D)
    elif layout == "column_scale":
        scale = (torch.rand(D, generator=gen) + 0.5).to(hip_device)
    elif layout == "dense_scale":
        scale = (torch.rand(B, K, D, generator=gen) + 0.5).to(hip_device)
    loc = loc.clone().requires_grad_(True) if layout == "dense" else loc
    dist = state.set_batch_shape_mode(torch.distributions.Normal(loc, scale, validate_args=False),
                                      state.BatchShapeMode.FULLY_EXPANDED)
    calls = {"drawn": 0}
    real = kernels.normal_rsample_drawn

    def spy(*args, **kwargs):
        calls["drawn"] += 1
        return real(*args, **kwargs)

    kernels.normal_rsample_drawn = spy
    try:
        torch.manual_seed(11)
        draw = state.sample(dist, B, K)
        after = torch.cuda.default_generators[hip_device.index or 0].get_offset()
    finally:
        kernels.normal_rsample_drawn = real
    expected = 1 if B * K * D >= kernels.RSAMPLE_DRAWN_MIN_ELEMENTS else 0      # (small draws keep the two launches)
    assert calls["drawn"] == expected and draw.shape == (B, K, D) and draw.is_contiguous()
    torch.manual_seed(11)
    eps = torch.empty(B, K, D, device=hip_device).normal_()
    assert torch.cuda.default_generators[hip_device.index or 0].get_offset() == after
    want = loc.detach() + eps * scale
    assert torch.equal(draw.detach(), want)
    if layout == "dense":
        weights = torch.randn(B, K, D, generator=gen).to(hip_device)
        (draw * weights).sum().backward()
        assert torch.equal(loc.grad, weights)
    # a scale that wants its gradient keeps the noise: the launch is not taken
    learned = torch.tensor(0.7, device=hip_device, requires_grad=True)
    kept = state.set_batch_shape_mode(torch.distributions.Normal(loc.detach(), learned, validate_args=False),
                                      state.BatchShapeMode.FULLY_EXPANDED)
    kernels.normal_rsample_drawn = spy
    try:
        torch.manual_seed(11)
        again = state.sample(kept, B, K)
    finally:
        kernels.normal_rsample_drawn = real
    assert calls["drawn"] == expected and torch.equal(again.detach(), loc.detach() + eps * learned.detach())
    again.sum().backward()
    assert float((learned.grad - eps.sum()).abs()) <= 1e-3 * max(1.0, float(eps.abs().sum()) ** 0.5)
