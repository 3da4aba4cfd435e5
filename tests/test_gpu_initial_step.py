"""K20 (aesmc_affine_normal_initial_step): the first timestep of a run — the proposal's transposed reparameterised draw
(aesmc/state.py:98, :102-103), the emission's location and the step's log-weight (aesmc/inference.py:79-98) for the
reference's model style (test/models/lgssm.py) — in one launch, against the three launches it stands for (K6 the
draw, K8 the location, K5 the log-weight: each pinned to `oracle/` by its own tests), BIT FOR BIT; against the oracle's
float64 statement of the same step; and through `infer` / `get_loss`, with and without gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


def _operands(B, K, dx, dy, device, seed, loc_q_rows, scale_q, loc_p_rows, scale_p, offset, scale_g, strided_weight):
    gen = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=device, generator=gen)
    pos = lambda *s: torch.rand(*s, device=device, generator=gen) * 1.5 + 0.25

    def param(kind, d, positive=False):
        make = pos if positive else r
        if kind == "scalar":
            return make(1).reshape(()).clone()
        if kind == "vector":
            return make(d)
        return make(B, d)          # "rows"

    o = {"eps": r(K, B, dx), "y": r(B, dy),
         "loc_q": param(loc_q_rows, dx), "scale_q": param(scale_q, dx, True),
         "loc_p": param(loc_p_rows, dx), "scale_p": param(scale_p, dx, True), "scale_g": param(scale_g, dy, True)}
    C = r(dy, dx) * 0.4
    o["C"] = C.t().contiguous().t() if strided_weight else C
    o["off"] = None if offset == "none" else param(offset, dy)
    return o


def _views(o, B, K, dx, dy):
    full = lambda t, d: (t if t.dim() < 2 else t.unsqueeze(1)).expand(B, K, d)
    return (full(o["loc_q"], dx), full(o["scale_q"], dx), full(o["loc_p"], dx), full(o["scale_p"], dx),
            full(o["y"], dy), full(o["scale_g"], dy))


def _three_launches(kernels, o, B, K, dx, dy):
    loc_q, scale_q, loc_p, scale_p, y, scale_g = _views(o, B, K, dx, dy)
    x = kernels.normal_rsample(o["eps"].transpose(0, 1), loc_q, scale_q)                  # K6, transposed
    loc_g = kernels.particle_affine(x, o["C"], o["off"])                                  # K8
    lw = kernels.normal_logweight(x, loc_p, scale_p, y, loc_g, scale_g.expand_as(loc_g), loc_q, scale_q)      # K5
    assert lw is not None
    return x, lw


CASES = [
    # B, K, dx, dy, loc_q, scale_q, loc_p, scale_p, offset, scale_g, strided C
    (64, 1024, 10, 10, "rows", "scalar", "vector", "vector", "none", "scalar", False),      # the bench model's first step
    (16, 32, 10, 10, "rows", "scalar", "vector", "vector", "none", "scalar", False),        # exactly one tile
    (17, 33, 10, 10, "rows", "scalar", "vector", "vector", "vector", "scalar", False),      # one row / particle past a tile
    (3, 1000, 1, 1, "rows", "scalar", "scalar", "scalar", "none", "scalar", False),         # the reference's 1-D LGSSM
    (5, 77, 3, 11, "vector", "vector", "rows", "rows", "rows", "vector", True),
    (40, 130, 16, 16, "rows", "rows", "scalar", "scalar", "vector", "rows", False),
    (9, 513, 4, 1, "scalar", "scalar", "scalar", "vector", "none", "scalar", True),
    (2, 4096, 7, 13, "rows", "vector", "vector", "scalar", "rows", "scalar", False),
    (1, 1, 2, 2, "rows", "scalar", "vector", "vector", "none", "scalar", False),
    (130, 64, 12, 5, "rows", "scalar", "vector", "vector", "vector", "vector", False),
]


@pytest.mark.parametrize("case", CASES)
def test_one_launch_gives_the_three_launches_bits(kernels, hip_device, case):
    B, K, dx, dy = case[:4]
    o = _operands(B, K, dx, dy, hip_device, 7 + B + K, *case[4:])
    want_x, want_lw = _three_launches(kernels, o, B, K, dx, dy)
    got_x = torch.full((B, K, dx), float("nan"), device=hip_device)
    got_lw = kernels.affine_initial_step(o["eps"], *_views(o, B, K, dx, dy)[:4], _views(o, B, K, dx, dy)[4], o["C"], o["off"],
                                         _views(o, B, K, dx, dy)[5], got_x)
    assert got_lw is not None
    assert torch.equal(got_x, want_x)
    assert got_lw.cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes()
    # and the oracle's statement of the step in float64 (the draw exactly: one rounded product, one rounded sum)
    f64 = {k: (v.double().cpu().numpy() if torch.is_tensor(v) else v) for k, v in o.items()}
    row = lambda v, d: np.broadcast_to(v if v.ndim < 2 else v[:, None, :], (B, K, d))
    eps = np.transpose(o["eps"].cpu().numpy(), (1, 0, 2))
    x32 = (row(o["loc_q"].cpu().numpy(), dx) + (eps * row(o["scale_q"].cpu().numpy(), dx)).astype(np.float32)).astype(np.float32)
    np.testing.assert_array_equal(got_x.cpu().numpy(), x32)
    x = x32.astype(np.float64)
    loc_g = x @ f64["C"].T + (0.0 if o["off"] is None else row(f64["off"], dy))

    def logn(v, mu, s):
        return (-((v - mu) ** 2) / (2 * s * s) - np.log(s) - 0.5 * np.log(2 * np.pi)).sum(-1)

    want = logn(x, row(f64["loc_p"], dx), row(f64["scale_p"], dx)) + \
        logn(row(f64["y"], dy), loc_g, row(f64["scale_g"], dy)) - logn(x, row(f64["loc_q"], dx), row(f64["scale_q"], dx))
    np.testing.assert_allclose(got_lw.cpu().numpy(), want, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want).max())))


def test_what_the_launch_declines(kernels, hip_device):
    B, K, dx, dy = 8, 64, 10, 10
    o = _operands(B, K, dx, dy, hip_device, 3, "rows", "scalar", "vector", "vector", "none", "scalar", False)
    v = list(_views(o, B, K, dx, dy))
    out = torch.empty(B, K, dx, device=hip_device)
    varying = torch.randn(B, K, dx, device=hip_device)                # a location that varies along the particles
    assert kernels.affine_initial_step(o["eps"], varying, v[1], v[2], v[3], v[4], o["C"], o["off"], v[5], out) is None
    assert kernels.affine_initial_step(o["eps"].double(), *v[:4], v[4], o["C"], o["off"], v[5], out) is None
    assert kernels.affine_initial_step(o["eps"].transpose(0, 1), *v[:4], v[4], o["C"], o["off"], v[5], out) is None
    wide = torch.randn(17, dx, device=hip_device)
    assert kernels.affine_initial_step(o["eps"], *v[:4], v[4].expand(B, K, dy), wide, None, v[5], out) is None


def _run(model, observations, K, algorithm, grad, seed=3):
    from aesmc_amd import inference, losses
    np.random.seed(seed)
    torch.manual_seed(seed)
    parts = (model.initial, model.transition, model.emission, model.proposal)
    if grad:
        for p in model.parameters():
            p.grad = None
        loss = losses.get_loss(observations, K, algorithm, *parts)
        loss.backward()
        return loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad():
        out = inference.infer("smc" if algorithm == "aesmc" else "is", observations, *parts, K,
                              return_log_marginal_likelihood=True, return_latents=True, return_log_weights=True,
                              return_ancestral_indices=algorithm == "aesmc")
    return out


@pytest.mark.parametrize("kind", ["affine", "matmul", "reference1d", "learned_scale"])
def test_a_run_through_the_one_launch_is_the_run_through_the_three(hip_device, kernels, kind):
    """`infer` / `get_loss` with `settings.initial_step` on and off: every returned value and every parameter gradient
    identical, bit for bit — the AffineNormal model, the same model in the reference's `Normal(x @ W.t() + c, s)` style, the
    reference's own 1-D classes (test/models/lgssm.py) and a model with learned scales; and the first timestep is then one
    launch (no K6, no K8, no K5)."""
    from aesmc_amd import settings
    from aesmc_amd.testing import models
    B, K, T = 6, 700, 4
    if kind == "reference1d":
        model = models.ReferenceLgssm1d().to(hip_device)
        observations = [o.to(hip_device) for o in model.simulate(T, B, seed=1)]
    elif kind == "learned_scale":
        model = models.LearnedScaleSsm(6, seed=0, validate_args=False).to(hip_device)
        observations = model.simulate(T, B, seed=1)
    else:
        model = models.LgssmNd(10, seed=0, affine=kind == "affine").to(hip_device).tune_proposal()
        observations = [o.to(hip_device) for o in model.simulate(T, B, seed=1)]
    counted = ("affine_initial_step", "normal_rsample", "normal_logweight")
    calls = dict.fromkeys(counted, 0)
    originals = {name: getattr(kernels, name) for name in counted}
    for name in counted:
        def spy(*args, _name=name, **kwargs):
            out = originals[_name](*args, **kwargs)
            calls[_name] += out is not None
            return out
        setattr(kernels, name, spy)
    exact = kind == "affine"      # (the other kinds state a location as `x @ C.t()`: at the first step the three-launch
    #                               route hands that to PyTorch's GEMM, the one-launch route records it — K8's chain, K11's
    #                               outer sums: the same numbers to float32 rounding of another summation order)

    def same(a, b, what):
        if exact or a.dtype == torch.int64:
            assert torch.equal(a, b), what
        else:
            scale = max(float(b.abs().max()), 1e-30)
            assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)

    try:
        for algorithm in ("aesmc", "iwae"):
            with settings.override(initial_step=False):
                for name in counted:
                    calls[name] = 0
                want = _run(model, observations, K, algorithm, grad=False)
                three = dict(calls)
                want_loss, want_grads = _run(model, observations, K, algorithm, grad=True)
            for name in counted:
                calls[name] = 0
            got = _run(model, observations, K, algorithm, grad=False)
            one = dict(calls)
            got_loss, got_grads = _run(model, observations, K, algorithm, grad=True)
            for key in ("log_marginal_likelihood", "log_weight"):
                same(got[key], want[key], (algorithm, key))
            for key in ("latents", "log_weights") + (("ancestral_indices",) if algorithm == "aesmc" and exact else ()):
                assert len(got[key]) == len(want[key])
                for a, b in zip(got[key], want[key]):
                    same(a, b, (algorithm, key))
            same(got_loss, want_loss, algorithm)
            assert sorted(got_grads) == sorted(want_grads) and len(got_grads) > 0
            for name in want_grads:
                same(got_grads[name], want_grads[name], (algorithm, name))
            if kind in ("affine", "matmul"):      # the first timestep: one launch where there were K6 and K5 (and K8)
                assert one["affine_initial_step"] == 1 and three["affine_initial_step"] == 0, (one, three)
                assert one["normal_rsample"] == three["normal_rsample"] - 1, (one, three)
                assert one["normal_logweight"] == three["normal_logweight"] - 1, (one, three)
            elif kind == "reference1d":       # [B, K] latents without a trailing extent: the three launches
                assert one["affine_initial_step"] == 0, one
            # ('iwae' differentiates the log-weights themselves: with gradients the three launches' own nodes take the step)
    finally:
        for name, fn in originals.items():
            setattr(kernels, name, fn)


def test_a_model_that_reads_the_first_draw_gets_its_values(hip_device, kernels):
    """The first draw is lazy only for as long as nobody looks: an emission that does arithmetic on `latents[-1]` the
    launch does not know (here a tanh) gets K6's transposed draw on the spot — the numbers of the eager route."""
    from aesmc_amd import settings
    from aesmc_amd.testing import models

    class Squashed(models.LgssmNd):
        def emission(self, latents=None, time=None, previous_observations=None):
            loc = torch.tanh(latents[-1]) @ self.C.t()
            return self._tag(self._normal(loc, self.emission_scale), "FULLY_EXPANDED")

    model = Squashed(4, seed=0).to(hip_device)
    observations = [o.to(hip_device) for o in model.simulate(3, 5, seed=1)]
    with settings.override(initial_step=False):
        want = _run(model, observations, 300, "aesmc", grad=False)
        want_loss, want_grads = _run(model, observations, 300, "aesmc", grad=True)
    got = _run(model, observations, 300, "aesmc", grad=False)
    got_loss, got_grads = _run(model, observations, 300, "aesmc", grad=True)
    assert torch.equal(got["log_marginal_likelihood"], want["log_marginal_likelihood"])
    for a, b in zip(got["latents"], want["latents"]):
        assert torch.equal(a, b)
    assert torch.equal(got_loss, want_loss)
    for name in want_grads:
        assert torch.equal(got_grads[name], want_grads[name]), name


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("through_lse", [True, False])
def test_the_first_steps_backward_rows_form_gives_the_generic_kernels_bits(kernels, hip_device, case, through_lse):
    """K5's backward for the first timestep's layout — x and the emission's location dense, every other operand constant
    along the particles — against the generic strided kernel (forced by handing x in as a strided view of the same
    values): every gradient it writes, bit for bit; with K1's softmax term formed in place and with a given grad_lw."""
    B, K, dx, dy = case[:4]
    if K < 2:
        pytest.skip("one particle per row: the generic kernel")
    o = _operands(B, K, dx, dy, hip_device, 11 + B + K, *case[4:])
    loc_q, scale_q, loc_p, scale_p, y, scale_g = _views(o, B, K, dx, dy)
    x, lw = _three_launches(kernels, o, B, K, dx, dy)
    loc_g = kernels.particle_affine(x, o["C"], o["off"])
    gen = torch.Generator(device=hip_device).manual_seed(5)
    need = [True] * 8
    if through_lse:
        lse = torch.logsumexp(lw, dim=1)
        extra = dict(lw=lw, lse=lse, grad_lse=torch.randn(B, device=hip_device, generator=gen))
        grad_lw = None
    else:
        extra, grad_lw = {}, torch.randn(B, K, device=hip_device, generator=gen)
    got = kernels.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad_lw, need, **extra)
    rows_form = (K * dx) % 4 == 0 and (K * dy) % 4 == 0      # a batch row of whole 16-byte vectors
    assert kernels._lib.aesmc_test_last_logweight_backward_form() == (2 if rows_form else 3)
    wider = torch.zeros(B, K, dx + 1, device=hip_device)
    wider[:, :, :dx] = x
    want = kernels.normal_logweight_backward(wider[:, :, :dx], loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad_lw,
                                             need, **extra)
    assert kernels._lib.aesmc_test_last_logweight_backward_form() == 3
    assert len(got) == len(want) == 8
    for a, b in zip(got, want):
        assert a is not None and b is not None and torch.equal(a, b)
