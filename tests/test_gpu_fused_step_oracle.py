"""The fused propagation step against `oracle/` first hand, configs[3] at its own size, the forms of the step's backward,
the wide (d = 128) step forward and backward (added in rounds 4 and 5):

  * the propagation launch that draws its own noise AND fetches x_{t-1} through the ancestors — the kernel that
    takes most of the forward pass — against `oracle/` FIRST HAND: the C restatement's gather, fma-chain draw and
    log-weight on the host, fed the noise `torch.empty(shape).normal_()` holds for the same generator state, at
    the north-star shape (1024, 4096, 10) and at ragged shapes.  Until now its parity was transitive (equal to
    other product kernels which equal the oracle);
  * BASELINE.json configs[3] (nonlinear SSM + MLP proposal) at its own per-GPU size B=128, K=4096, T=100:
    size-independent properties (finite log Z, sorted in-range ancestors, finite loss and gradients) and a sample
    of batch rows against the CPU port of the reference run on those rows alone;
  * the two forms of the fused propagation launch (scalar-register weights / matrix cores) and of the step's backward
    (rows in registers / tiles through LDS) against each other, bit for bit;
  * configs[4]'s extent (rows of 128 values): the step on the fp32 matrix cores (K17 + K18) against the C oracle —
    x_t bit for bit at (2, 16384, 128) —, its in-kernel noise against `torch.empty(...).normal_()`, and `infer` end to
    end against the GEMM route.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from tests.test_gpu_linear_gaussian import operands
from tests.test_gpu_noise_and_lazy_latents import _ancestors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


ORACLE_SHAPES = [(1024, 4096, 10, 10), (300, 4099, 10, 7), (37, 29000, 6, 9), (1, 128, 2, 2), (2, 4096, 12, 12),
                 (5, 777, 3, 11), (64, 1024, 8, 4), (3, 200000, 10, 10)]


@pytest.mark.parametrize("spread", [1.0, 5.0])
@pytest.mark.parametrize("shape", ORACLE_SHAPES)
def test_propagation_with_noise_and_gather_inside_equals_the_c_oracle(kernels, hip_device, shape, spread, monkeypatch):
    """aesmc_affine_normal_propagate_drawn against oracle/smc_core.c: x_t bit for bit (gather of the ancestor rows,
    one fma chain per element started from the offset, eps * s rounded before the sum), the log-weight to the
    tolerance the stand-alone log-weight kernel is held to against the same C function (the device's log(sigma)
    against glibc's, times d: 5e-7 relative).  spread 1: a healthy ancestry, 5: a collapsed one."""
    from aesmc_amd import _philox
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)
    B, K, dx, dy = shape
    n, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=B + K)       # the maps and the scales
    gen = torch.Generator(device=hip_device).manual_seed(K + dx)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    off_p = torch.randn(dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=spread)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], off_q))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    torch.manual_seed(77 + K)
    torch.randn(5, device=hip_device)
    state = torch.cuda.get_rng_state(hip_device)
    eps = torch.empty(B, K, dx, device=hip_device).normal_()      # what the reference's rsample would draw (state.py:98)
    torch.cuda.set_rng_state(state, hip_device)
    reservation = _philox.reserve(B * K * dx, hip_device)
    got_x = torch.full_like(x_prev, float("nan"))
    got_lw = kernels.affine_propagate_drawn(x_prev, reservation, y, *terms, scales, out_x=got_x, ancestors=idx)
    assert got_lw is not None
    assert kernels.read_flags(hip_device) == 0
    # ---- the oracle, on the host
    moved, flags = c_oracle.gather(x_prev.cpu().numpy(), idx.cpu().numpy())
    assert flags == 0
    s_p, s_g, s_q = (float(s.cpu()) for s in scales)
    want_x = c_oracle.affine_rsample(moved, n["Q"], off_q.cpu().numpy(), eps.cpu().numpy(), s_q)
    np.testing.assert_array_equal(got_x.cpu().numpy(), want_x)
    want_lw = c_oracle.affine_logweight(moved, want_x, y.cpu().numpy(), (n["A"], off_p.cpu().numpy()),
                                        (n["C"], n["off_g"]), (n["Q"], off_q.cpu().numpy()), s_p, s_g, s_q)
    np.testing.assert_allclose(got_lw.cpu().numpy(), want_lw, rtol=5e-7,
                               atol=5e-7 * max(1.0, float(np.abs(want_lw).max())))


def test_configs3_nonlinear_model_at_its_own_size(hip_device):
    """BASELINE.json configs[3] on one GPU's shard (B=128, K=4096, T=100, d=10, MLP proposal): properties that do
    not depend on the size, then 4 batch rows over the first 5 timesteps against the CPU port of the reference
    (`oracle/reference_port.py`) replaying the same draws.  float32 tolerances as in DESIGN.md section 5."""
    from aesmc_amd import inference, losses, state
    from aesmc_amd.testing import models, replay
    from oracle import reference_port
    B, K, T, d = 128, 4096, 100, 10
    model = models.NonlinearSsm(d, hidden=64, seed=0, dtype=torch.float32, state=state, fused=True).to(hip_device)
    observations = model.simulate(T, B, seed=1)
    np.random.seed(11)
    torch.manual_seed(11)
    with torch.no_grad():
        result = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal, K,
                                 return_log_marginal_likelihood=True, return_ancestral_indices=True,
                                 return_latents=False, return_log_weight=False)
    log_z = result["log_marginal_likelihood"]
    assert log_z.shape == (B,) and bool(torch.isfinite(log_z).all())
    indices = result["ancestral_indices"]
    assert len(indices) == T - 1
    for a in indices[::7] + indices[-1:]:
        assert a.shape == (B, K) and a.dtype == torch.int64
        assert int(a.min()) >= 0 and int(a.max()) < K
        assert bool((a[:, 1:] >= a[:, :-1]).all())                 # systematic resampling: non-decreasing along k
    del result, indices
    np.random.seed(12)
    torch.manual_seed(12)
    loss = losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
    assert bool(torch.isfinite(loss))
    loss.backward()
    grads = {name: p.grad for name, p in model.named_parameters() if p.grad is not None}
    assert {"A", "C"} <= set(grads) and any(name.startswith("net.") for name in grads)
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())
    assert all(float(g.abs().max()) > 0 for g in grads.values())
    del loss, grads
    # ---- a sample of rows against the port of the reference, first 5 timesteps, draws replayed
    rows, Ts = [0, 37, 90, 127], 5
    cpu_model = models.NonlinearSsm(d, hidden=64, seed=0, dtype=torch.float32, state=reference_port)
    cpu_obs = [o[rows].cpu() for o in observations[:Ts]]
    np.random.seed(5)
    torch.manual_seed(5)
    flags = dict(return_log_marginal_likelihood=True, return_log_weights=True, return_ancestral_indices=True)
    with replay.record() as tape:
        want = reference_port.infer("smc", cpu_obs, cpu_model.initial, cpu_model.transition, cpu_model.emission,
                                    cpu_model.proposal, K, **flags)
    gpu_obs = [o.to(hip_device) for o in cpu_obs]
    with replay.replay(tape), torch.no_grad():
        got = inference.infer("smc", gpu_obs, model.initial, model.transition, model.emission, model.proposal, K, **flags)
    first = float((got["log_weights"][0].cpu() - want["log_weights"][0]).abs().max())
    assert first < 2e-4, first
    # K = 4096 in float32: a float64-CDF resampler flips ~1e-3 of the reference's float32-CDF indices by one
    # (SURVEY section 7), after which the two runs hold different particles in those slots: the FIRST resampling
    # step is held to that rate, the later ones only through log Z
    first_agree = float((got["ancestral_indices"][0].cpu() == want["ancestral_indices"][0]).double().mean())
    reference = want["log_marginal_likelihood"].double()
    err = float(((got["log_marginal_likelihood"].cpu().double() - reference).abs() / (1 + reference.abs())).max())
    assert first_agree >= 0.995 and err < 1e-2, (first_agree, err)


@pytest.mark.parametrize("shape", [(9, 4099, 10, 10), (3, 20000, 10, 10), (4, 9000, 7, 5)])
def test_both_forms_of_the_fused_launch_give_the_same_bits(kernels, hip_device, shape, monkeypatch):
    """The fused propagation launch evaluates the three maps on the vector ALU with scalar-register weights when the
    rows of every weight are contiguous, and on the matrix cores otherwise (strided views, extents it has no
    exact instantiation for).  Same chains, same order: the two forms and the unfused route (gather, torch's
    normal_, K15) agree bit for bit."""
    from aesmc_amd import _philox
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)
    B, K, dx, dy = shape
    n, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=K)
    gen = torch.Generator(device=hip_device).manual_seed(K)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=K, spread=1.0)
    scales = (o["s_p"], o["s_g"], o["s_q"])
    strided = {name: o[name].t().contiguous().t() for name in ("A", "C", "Q")}       # same values, columns contiguous
    assert all(not strided[name].is_contiguous() or strided[name].shape[0] == 1 for name in strided)
    results = []
    for weights in (o, strided):
        terms = ((weights["A"], None), (weights["C"], o["off_g"]), (weights["Q"], off_q))
        torch.manual_seed(5)
        state = torch.cuda.get_rng_state(hip_device)
        eps = torch.empty(B, K, dx, device=hip_device).normal_()
        torch.cuda.set_rng_state(state, hip_device)
        reservation = _philox.reserve(B * K * dx, hip_device)
        got_x = torch.full_like(x_prev, float("nan"))
        got_lw = kernels.affine_propagate_drawn(x_prev, reservation, y, *terms, scales, out_x=got_x, ancestors=idx)
        assert got_lw is not None
        results.append((got_x, got_lw))
    moved = kernels.gather(x_prev, idx)
    want_x = torch.full_like(moved, float("nan"))
    want_lw = kernels.affine_propagate(moved, eps, y, (o["A"], None), (o["C"], o["off_g"]), (o["Q"], off_q), scales,
                                       out_x=want_x)
    for got_x, got_lw in results:
        assert torch.equal(got_x, want_x)
        assert torch.equal(got_lw, want_lw)
    assert kernels.read_flags(hip_device) == 0


@pytest.mark.parametrize("strided", [False, True])
def test_the_fused_launch_flags_and_survives_ancestors_out_of_range(kernels, hip_device, strided, monkeypatch):
    """Ancestors that torch.gather would reject — K (what the resampling launch writes for a degenerate row), a
    negative one, one beyond 2^32 — are flagged (AESMC_FLAG_INDEX_OUT_OF_RANGE, raised as a RuntimeError by `infer`)
    and clamped: no fault, and every particle with a valid ancestor gets exactly the value of a clean run."""
    from aesmc_amd import _philox
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)
    B, K, dx, dy = 5, 4096, 10, 10
    n, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=2)
    weights = {name: (o[name].t().contiguous().t() if strided else o[name]) for name in ("A", "C", "Q")}
    gen = torch.Generator(device=hip_device).manual_seed(9)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    terms = ((weights["A"], None), (weights["C"], o["off_g"]), (weights["Q"], None))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    idx = _ancestors(B, K, hip_device, seed=4, spread=1.0)
    bad = idx.clone()
    bad[1, 17] = K
    bad[2, 4000] = -1
    bad[4, 63] = (1 << 33) + 5
    outs = []
    for ancestors in (idx, bad):
        torch.manual_seed(8)
        reservation = _philox.reserve(B * K * dx, hip_device)
        out_x = torch.full_like(x_prev, float("nan"))
        lw = kernels.affine_propagate_drawn(x_prev, reservation, y, *terms, scales, out_x=out_x, ancestors=ancestors)
        assert lw is not None
        outs.append((out_x, lw, kernels.read_flags(hip_device)))
    (clean_x, clean_lw, clean_flags), (bad_x, bad_lw, bad_flags) = outs
    assert clean_flags == 0 and bad_flags != 0
    assert bool(torch.isfinite(bad_x).all()) and bool(torch.isfinite(bad_lw).all())
    valid = torch.ones(B, K, dtype=torch.bool, device=hip_device)
    valid[1, 17] = valid[2, 4000] = valid[4, 63] = False
    assert torch.equal(bad_x[valid], clean_x[valid]) and torch.equal(bad_lw[valid], clean_lw[valid])


# ---- K14's second form (rows in registers, scalar weights) against its first, bit for bit ---------
def _step_backward_both_forms(kernels, call, grid=0):
    """`call()` under the first form (tiles through LDS) and under the second, both on `grid` workgroups (0: each
    form's own); every gradient returned by both, as bytes."""
    lib = kernels._lib
    results = []
    try:
        for form in (1, 0):
            assert lib.aesmc_test_set_step_backward(form, grid) == 0
            grads = call()
            torch.cuda.synchronize()
            # (the comparison must be between the two kernels: 1 = tiles through LDS, 2 = rows in registers)
            assert lib.aesmc_test_last_step_backward_form() == (1 if form == 1 else 2), "the rows form declined the call"
            results.append([None if g is None else g.detach().cpu().numpy().copy() for g in grads])
    finally:
        lib.aesmc_test_set_step_backward(0, 0)
    return results


# (B, K, d): rows of ten values at the sizes the form was built on, then every other even extent it takes
STEP_BACKWARD_SHAPES = [(1024, 4096, 10), (3, 256, 10), (5, 1024, 10), (130, 512, 10), (1, 256, 10), (2, 8192, 10),
                        (64, 4096, 4), (3, 256, 2), (5, 1024, 6), (130, 512, 8), (2, 8192, 12), (7, 768, 14), (33, 1024, 8),
                        (4, 2048, 4)]


@pytest.mark.parametrize("arrives", ["nothing", "grad_x", "children", "children_collapsed", "children_and_grad_x"])
@pytest.mark.parametrize("shape", STEP_BACKWARD_SHAPES)
def test_both_forms_of_the_step_backward_give_the_same_bits(kernels, hip_device, shape, arrives):
    """aesmc_affine_step_backward_resampled for rows of an even number (2 .. 14) of float32 values: the form that keeps a wavefront's rows in
    registers and reads the weights as scalar operands (linear_gaussian_step_backward.hip) equals the form that stages
    tiles through LDS in every output bit — particle gradients, the three weight gradients, the offsets' row sums, the
    scales — on the same grid (the records' association follows the grid), with and without the gather's backward
    folded in, for a healthy and a collapsed next-step ancestry (runs longer than a lane sums by itself)."""
    from tests.test_gpu_noise_and_lazy_latents import _next_resampling
    B, K, d = shape
    _, o = operands(B, K, d, d, np.float32, hip_device, seed=B + K)
    off_p = torch.from_numpy(np.random.RandomState(4).randn(d).astype(np.float32)).to(hip_device)
    terms = ((o["A"], off_p if B % 2 else None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    anc = _ancestors(B, K, hip_device, seed=3, spread=1.0)
    moved = kernels.gather(o["x_prev"], anc)
    x = kernels.affine_rsample(moved, o["Q"], o["off_q"], o["eps"], o["s_q"])
    lw = kernels.affine_logweight(moved, x, o["y"], *terms, scales)
    lse = kernels.logweight_lse(lw, None, None, want_lw=False)[1]
    rng = np.random.RandomState(B)
    glse = torch.from_numpy(rng.randn(B).astype(np.float32)).to(hip_device)
    extra = {"ancestors": anc}
    if arrives in ("grad_x", "children_and_grad_x"):
        extra["grad_x"] = torch.from_numpy(rng.randn(B, K, d).astype(np.float32)).to(hip_device)
    if arrives.startswith("children"):
        _, child_end = _next_resampling(kernels, B, K, hip_device, seed=7 * B + K,
                                        spread=6.0 if arrives == "children_collapsed" else 1.0)
        extra["child_grad"] = torch.from_numpy(rng.randn(B, K, d).astype(np.float32)).to(hip_device)
        extra["child_end"] = child_end
    need = [True, False, False, True, terms[0][1] is not None, True, True, True, True, True, True, True]
    call = lambda: kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=glse, **extra)
    tiles = B * K // 256
    first, second = _step_backward_both_forms(kernels, call, grid=min(tiles, 768))
    assert len(first) == len(second)
    for slot, (a, b) in enumerate(zip(first, second)):
        assert (a is None) == (b is None), slot
        if a is None:
            continue
        if d == 10 or slot < 3:
            # rows of ten values: both forms give a lane ONE particle, so every sum associates alike — every bit;
            # particle gradients are per-particle chains at any extent
            assert a.tobytes() == b.tobytes(), (slot, float(np.abs(a - b).max()))
        else:
            # other extents: the tiles form gives a lane two particles where its registers allow, so the sums over
            # particles (weights, offsets, scales) associate differently: equal to rounding
            scale = max(float(np.abs(a).max()), 1e-30)
            assert float(np.abs(a - b).max()) <= 2e-5 * scale, (slot, float(np.abs(a - b).max()), scale)
    # and on its own grid: the same particle gradients, the sums to rounding
    own = [None if g is None else g.detach().cpu().numpy() for g in call()]
    np.testing.assert_array_equal(own[0], first[0])
    for slot in range(3, 12):
        if own[slot] is not None:
            scale = max(float(np.abs(first[slot]).max()), 1e-30)
            assert float(np.abs(own[slot] - first[slot]).max()) <= 2e-5 * scale, slot


# ---- configs[4]'s extent (rows of 128 values) on the fp32 matrix cores: K17 + K18 -----------------
@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("shape", [(1, 32), (2, 64), (3, 320), (5, 1024), (2, 16384)])
def test_the_wide_step_equals_the_c_oracle(kernels, hip_device, shape, gather):
    """aesmc_affine_normal_propagate_wide against oracle/smc_core.c at d = 128: x_t bit for bit (the gather of the
    ancestor rows, one fma chain per element started from the offset — 32 matrix-core k-steps —, eps * s rounded
    before the sum), the log-weight to 2e-6 relative (the squared distances are summed in another order than the
    oracle's single chain: three sums of 128 terms each ~ 1e2 .. 1e4).  Offsets: per batch row, shared, absent."""
    B, K = shape
    d = 128
    rng = np.random.RandomState(B * K)
    r = lambda *s: rng.randn(*s).astype(np.float32)
    host = {"x_prev": r(B, K, d), "eps": r(B, K, d), "y": r(B, d),
            "A": (0.9 * np.eye(d) + 0.05 * rng.randn(d, d)).astype(np.float32),
            "Q": (0.45 * np.eye(d) + 0.05 * rng.randn(d, d)).astype(np.float32),
            "C": (0.1 * rng.randn(d, d)).astype(np.float32), "off_q": r(B, d), "off_g": r(d)}
    o = {key: torch.from_numpy(value).to(hip_device) for key, value in host.items()}
    s_p, s_g, s_q = (torch.tensor(v, dtype=torch.float32, device=hip_device) for v in (1.0, 0.5, 0.7))
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    anc = _ancestors(B, K, hip_device, seed=B + K, spread=2.0) if gather else None
    out_x = torch.empty(B, K, d, device=hip_device)
    kernels.read_flags(hip_device)
    lw = kernels.affine_propagate_wide(o["x_prev"], o["eps"], o["y"], *terms, (s_p, s_g, s_q), out_x, ancestors=anc)
    assert lw is not None, "the wide launch declined the shape it is built for"
    moved = host["x_prev"] if anc is None else c_oracle.gather(host["x_prev"], anc.cpu().numpy())[0]
    want_x = c_oracle.affine_rsample(moved, host["Q"], host["off_q"], host["eps"], 0.7)
    np.testing.assert_array_equal(out_x.cpu().numpy(), want_x)
    want_lw = c_oracle.affine_logweight(moved, want_x, host["y"], (host["A"], None), (host["C"], host["off_g"]),
                                        (host["Q"], host["off_q"]), 1.0, 0.5, 0.7)
    got = lw.cpu().numpy()
    assert np.isfinite(got).all()
    scale = np.maximum(np.abs(want_lw), 1.0)
    assert float((np.abs(got - want_lw) / scale).max()) <= 2e-6 * 4
    assert kernels.read_flags(hip_device) == 0


# ---- every other width: K17g + K18g (aesmc/state.py:61-183 is dimension-agnostic) ----------------------
WIDTHS = [(24, 24), (32, 32), (64, 64), (96, 96), (256, 256), (32, 48), (192, 80), (48, 20), (128, 128), (20, 4), (100, 136),
          (160, 256), (17, 17), (18, 23), (19, 1), (33, 65), (127, 129), (255, 254), (130, 3)]


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("shape", [(3, 40), (2, 320), (1, 1000)])
@pytest.mark.parametrize("widths", WIDTHS)
def test_the_matrix_core_step_at_every_width_equals_the_c_oracle(kernels, hip_device, widths, shape, gather):
    """aesmc_affine_normal_propagate_wide at latent width dx (17 .. 256) and observation width dy (1 .. 256) — dx != dy
    included, padded inside the launch to 32 / 48 / 64 / 96 / 128 / 192 / 256, rows wider than 128 cut into chunks of
    output rows, widths that are not multiples of 4 moved element by element — and K not a multiple of 32 (a masked last
    tile) against oracle/smc_core.c: x_t bit for bit (one fma
    chain per element started from the offset; zero padding leaves a chain as it is), the log-weight to rounding (the
    squared distances are summed per lane, per particle and per chunk instead of in one chain).  (128, 128) at these K is
    the generic kernel too: K = 40 / 1000 are not whole tiles."""
    dx, dy = widths
    B, K = shape
    rng = np.random.RandomState(B * K + dx + 7 * dy)
    r = lambda *s: rng.randn(*s).astype(np.float32)
    host = {"x_prev": r(B, K, dx), "eps": r(B, K, dx), "y": r(B, dy),
            "A": (0.9 * np.eye(dx) + 0.3 / np.sqrt(dx) * rng.randn(dx, dx)).astype(np.float32),
            "Q": (0.45 * np.eye(dx) + 0.3 / np.sqrt(dx) * rng.randn(dx, dx)).astype(np.float32),
            "C": (1.0 / np.sqrt(dx) * rng.randn(dy, dx)).astype(np.float32), "off_q": r(B, dx), "off_g": r(dy)}
    o = {key: torch.from_numpy(value).to(hip_device) for key, value in host.items()}
    s_p, s_g, s_q = (torch.tensor(v, dtype=torch.float32, device=hip_device) for v in (1.0, 0.5, 0.7))
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    for (weight, offset) in terms:
        source = o["x_prev"]
        assert kernels.affine_wide_covers(source, weight, offset, s_p)
    anc = _ancestors(B, K, hip_device, seed=B + K, spread=2.0) if gather else None
    out_x = torch.full((B, K, dx), float("nan"), device=hip_device)
    kernels.read_flags(hip_device)
    lw = kernels.affine_propagate_wide(o["x_prev"], o["eps"], o["y"], *terms, (s_p, s_g, s_q), out_x, ancestors=anc)
    assert lw is not None, "the matrix-core step declined a width it is built for"
    moved = host["x_prev"] if anc is None else c_oracle.gather(host["x_prev"], anc.cpu().numpy())[0]
    want_x = c_oracle.affine_rsample(moved, host["Q"], host["off_q"], host["eps"], 0.7)
    np.testing.assert_array_equal(out_x.cpu().numpy(), want_x)
    want_lw = c_oracle.affine_logweight(moved, want_x, host["y"], (host["A"], None), (host["C"], host["off_g"]),
                                        (host["Q"], host["off_q"]), 1.0, 0.5, 0.7)
    got = lw.cpu().numpy()
    assert got.shape == (B, K) and np.isfinite(got).all()
    scale = np.maximum(np.abs(want_lw), 1.0)
    assert float((np.abs(got - want_lw) / scale).max()) <= 2e-6 * 4
    assert kernels.read_flags(hip_device) == 0


def test_widths_the_matrix_core_step_leaves_to_the_other_routes(kernels, hip_device):
    """At most 16 (the item kernels' side) or above 256: declined — the item kernels / the GEMM route apply."""
    g = torch.Generator(device=hip_device).manual_seed(0)
    one = torch.tensor(1.0, device=hip_device)
    for dx, dy in ((16, 16), (12, 40), (260, 260), (24, 260)):
        x = torch.randn(2, 64, dx, device=hip_device, generator=g)
        y = torch.randn(2, dy, device=hip_device, generator=g)
        A = torch.randn(dx, dx, device=hip_device, generator=g) * 0.05
        C = torch.randn(dy, dx, device=hip_device, generator=g) * 0.05
        assert kernels.affine_propagate_wide(x, x.clone(), y, (A, None), (C, None), (A, None), (one, one, one),
                                             torch.empty_like(x)) is None, (dx, dy)


def test_the_wide_step_declines_what_it_does_not_cover_and_survives_bad_ancestors(kernels, hip_device):
    d = 128
    g = torch.Generator(device=hip_device).manual_seed(0)
    x = torch.randn(2, 64, d, device=hip_device, generator=g)
    y = torch.randn(2, d, device=hip_device, generator=g)
    W = torch.randn(d, d, device=hip_device, generator=g) * 0.05
    one = torch.tensor(1.0, device=hip_device)
    out = torch.empty_like(x)
    terms = ((W, None), (W, None), (W, None))
    # (K = 40, not a multiple of 32, is covered since 0.5.0 — a masked last tile: tested with the other widths below)
    assert kernels.affine_propagate_wide(x, x, y, (W.t(), None), (W, None), (W, None), (one, one, one), out) is None
    # an observation K18 would misread: float64 bytes (torch.from_numpy data against a float32 model), another batch
    # extent, another device; offsets and scales of the wrong shape / dtype / device — declined, nothing launched
    before = out.zero_().clone()
    assert kernels.affine_propagate_wide(x, x, y.double(), *terms, (one, one, one), out) is None
    assert kernels.affine_propagate_wide(x, x, y[:1], *terms, (one, one, one), out) is None
    assert kernels.affine_propagate_wide(x, x, y.cpu(), *terms, (one, one, one), out) is None
    assert kernels.affine_propagate_wide(x, x, y, (W, y[0, :64]), (W, None), (W, None), (one, one, one), out) is None
    assert kernels.affine_propagate_wide(x, x, y, (W, y.double()), (W, None), (W, None), (one, one, one), out) is None
    assert kernels.affine_propagate_wide(x, x, y, *terms, (one, one.double(), one), out) is None
    assert kernels.affine_propagate_wide(x, x, y, *terms, (one, one, one.cpu()), out) is None
    assert torch.equal(out, before)
    anc = torch.zeros(2, 64, dtype=torch.int64, device=hip_device)
    anc[0, 3], anc[1, 5] = 64, -1
    kernels.read_flags(hip_device)
    lw = kernels.affine_propagate_wide(x, torch.zeros_like(x), y, *terms, (one, one, one), out, ancestors=anc)
    assert lw is not None and torch.isfinite(lw).all()
    assert kernels.read_flags(hip_device) != 0


def test_a_wide_model_runs_through_the_matrix_core_step_and_gives_the_other_routes_numbers(hip_device, monkeypatch):
    """configs[4]'s model (d = 128, AffineNormal callables) through `infer` under no_grad: every SMC step after the first
    is K17 + K18 (counted), the particles are the ones the GEMM + K6 + K5 route draws from the same seeds (x_t equal to
    1e-5: a library GEMM associates differently from the fma chain), log Z agrees to 1e-5 relative, and with gradients
    enabled the model takes the other routes untouched."""
    from aesmc_amd import _kernels, inference
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    model = LgssmNd(128, dtype=torch.float32, affine=True, validate_args=False, emission_scale=0.05).tune_proposal().to(hip_device)
    B, K, T = 3, 256, 5
    observations = model.simulate(T, B, seed=2)
    calls = {"wide": 0}
    real = provider.affine_propagate_wide

    def counting(*args, **kwargs):
        out = real(*args, **kwargs)
        calls["wide"] += out is not None
        return out
    monkeypatch.setattr(provider, "affine_propagate_wide", counting)

    def run():
        np.random.seed(4)
        torch.manual_seed(4)
        with torch.no_grad():
            return inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal, K,
                                   return_log_marginal_likelihood=True, return_latents=True, return_log_weight=True,
                                   return_original_latents=True)
    wide = run()
    assert calls["wide"] == T - 1
    monkeypatch.setattr(provider, "affine_wide_covers", lambda *a, **k: False)
    plain = run()
    assert calls["wide"] == T - 1
    # the first resampling sees identical weights; later steps' particles differ by the maps' rounding only as long as no
    # ancestor flips: compare the first two steps' draws, and the estimate as a whole
    for t in range(2):
        a, b = wide["original_latents"][t], plain["original_latents"][t]
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), t
    lml_w, lml_p = wide["log_marginal_likelihood"], plain["log_marginal_likelihood"]
    assert float(((lml_w - lml_p).abs() / lml_p.abs().clamp_min(1.0)).max()) <= 2e-3
    assert torch.isfinite(lml_w).all()
    monkeypatch.undo()
    # gradients enabled: not this route
    monkeypatch.setattr(provider, "affine_propagate_wide", counting)
    np.random.seed(4)
    torch.manual_seed(4)
    small = [o[:, :] for o in observations]
    out = inference.infer("smc", small, model.initial, model.transition, model.emission, model.proposal, 64,
                          return_log_marginal_likelihood=True)
    assert calls["wide"] == T - 1 and out["log_marginal_likelihood"].requires_grad


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("shape", [(1, 16384), (3, 16384), (2, 32768)])
def test_the_wide_step_forms_torchs_noise_itself(kernels, hip_device, shape, gather):
    """K17 handed a reservation instead of a noise tensor: x_t and the log-weights equal, bit for bit, the launch handed
    `torch.empty([B,K,128]).normal_()` drawn at the same generator state — the launch's 32-particle tiles take eight
    particles from each quarter of a Philox trip, two lanes share every call — and the generator ends where `normal_`
    leaves it.  Shapes the quarters do not fit (K not a multiple of 4 G / 128) are declined."""
    from aesmc_amd import _philox
    B, K = shape
    d = 128
    gen = torch.Generator(device=hip_device).manual_seed(B + K)
    make = lambda *s: torch.randn(*s, device=hip_device, generator=gen)
    x_prev, y, off_q = make(B, K, d), make(B, d), make(B, d)
    eye = torch.eye(d, device=hip_device)
    A, C, Q = 0.9 * eye + 0.05 * make(d, d), 0.1 * make(d, d), 0.45 * eye + 0.05 * make(d, d)
    scales = tuple(torch.tensor(v, device=hip_device) for v in (1.0, 0.5, 0.7))
    terms = ((A, None), (C, None), (Q, off_q))
    anc = _ancestors(B, K, hip_device, seed=B, spread=1.0) if gather else None
    torch.manual_seed(77)
    state = torch.cuda.get_rng_state(hip_device)
    eps = torch.empty(B, K, d, device=hip_device).normal_()
    after = torch.cuda.get_rng_state(hip_device)
    want_x = torch.empty_like(x_prev)
    want_lw = kernels.affine_propagate_wide(x_prev, eps, y, *terms, scales, want_x, ancestors=anc)
    torch.cuda.set_rng_state(state, hip_device)
    noise = _philox.reserve(B * K * d, hip_device)
    assert torch.equal(torch.cuda.get_rng_state(hip_device), after)
    got_x = torch.empty_like(x_prev)
    got_lw = kernels.affine_propagate_wide(x_prev, noise, y, *terms, scales, got_x, ancestors=anc)
    assert got_lw is not None and want_lw is not None
    assert torch.equal(got_x, want_x)
    assert got_lw.cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes()
    # a row shorter than a trip's four quarters: declined
    short = _philox.reserve(2 * 4096 * d, hip_device)
    assert kernels.affine_propagate_wide(x_prev[:2, :4096].contiguous(), short, y[:2], (A, None), (C, None), (Q, off_q[:2]),
                                         scales, torch.empty(2, 4096, d, device=hip_device)) is None if B >= 2 else True


# ---- the wide step under autograd: recomputation instead of retention ------------------------------------
@pytest.mark.parametrize("shape", [(2, 16384, 128, 128), (3, 1024, 128, 128), (2, 1000, 24, 24), (2, 512, 64, 48),
                                   (1, 320, 192, 80), (2, 96, 256, 256)])
def test_the_wide_steps_backward_equals_float64_autograd(kernels, hip_device, shape):
    """`affine_step_backward` on rows of 128 values (the backward of a K17 / K18 step, recomputed from x_{t-1}, the
    ancestors, x_t and the log-weights) against float64 autograd of the same step written with PyTorch operations —
    aesmc/state.py:114-155, :179 and aesmc/inference.py:108-130 for one timestep: every gradient (resampled rows, the
    three maps, their offsets — per batch row, shared, absent —, the observation, the three scales) to float32 rounding
    of sums over B K particles."""
    B, K, d, dy = shape      # (other widths than 128: K17g / K18g forward, the same recomputing backward)
    gen = torch.Generator(device=hip_device).manual_seed(B + K)
    make = lambda *s: torch.randn(*s, device=hip_device, generator=gen)
    x_prev, eps, y, off_q, off_g = make(B, K, d), make(B, K, d), make(B, dy), make(B, d), make(dy)
    eye = torch.eye(d, device=hip_device)
    A, C, Q = 0.9 * eye + 0.05 * make(d, d), 0.1 * make(dy, d), 0.45 * eye + 0.05 * make(d, d)
    scales = tuple(torch.tensor(v, device=hip_device) for v in (1.0, 0.5, 0.7))
    anc = _ancestors(B, K, hip_device, seed=B, spread=1.0)
    terms = ((A, None), (C, off_g), (Q, off_q))
    x = torch.empty_like(x_prev)
    lw = kernels.affine_propagate_wide(x_prev, eps, y, *terms, scales, x, ancestors=anc)
    assert lw is not None
    lse = torch.logsumexp(lw, dim=1)
    glse = make(B)
    grad_x = make(B, K, d) * 1e-3
    need = [True, False, True, True, False, True, True, True, True, True, True, True]
    got = kernels.affine_step_backward(x_prev, x, y, *terms, scales, need, lw, lse, grad_lse=glse, grad_x=grad_x, ancestors=anc)
    # float64 autograd of the same step
    f = lambda t: t.double().detach().clone().requires_grad_(True)
    moved = f(torch.gather(x_prev, 1, anc.unsqueeze(-1).expand_as(x_prev)))
    A_, C_, Q_, y_, og_, oq_ = f(A), f(C), f(Q), f(y), f(off_g), f(off_q)
    sp_, sg_, sq_ = (f(s_) for s_ in scales)
    loc_q = moved @ Q_.t() + oq_.unsqueeze(1)
    noise = ((x.double() - loc_q) / sq_).detach()      # the eps the launch used, as float64
    x_t = loc_q + sq_ * noise
    logn = lambda v, loc, sc: torch.distributions.Normal(loc, sc).log_prob(v).sum(2)
    lw64 = logn(x_t, moved @ A_.t(), sp_) + logn(y_.unsqueeze(1), x_t @ C_.t() + og_, sg_) - logn(x_t, loc_q, sq_)
    loss = (glse.double() * torch.logsumexp(lw64, dim=1)).sum() + (grad_x.double() * x_t).sum()
    loss.backward()
    want = {0: moved.grad, 2: y_.grad, 3: A_.grad, 5: C_.grad, 6: og_.grad, 7: Q_.grad, 8: oq_.grad, 9: sp_.grad, 10: sg_.grad,
            11: sq_.grad}
    for slot, reference in want.items():
        assert got[slot] is not None, slot
        scale = float(reference.abs().max()) + 1e-30
        error = float((got[slot].double().reshape(reference.shape) - reference).abs().max())
        assert error <= 5e-4 * scale, (slot, error, scale)
    assert got[1] is None and got[4] is None


def test_a_wide_model_trains_through_the_matrix_core_step(hip_device, monkeypatch):
    """configs[4]'s model (d = 128) through `get_loss` WITH gradients: every resampled step's forward is K17 + K18
    (counted), its backward the recomputing adjoint.  Loss and every parameter gradient agree with the GEMM route's
    autograd — run on the same seeds and handed the wide run's ancestors at every resampling (teacher forcing: at this
    extent the log-weights are of order 1e4, float32 rounding differs between two routes by ~1e-2 absolute, and two
    free-running evaluations resample a few per cent of their particles differently: both valid, not comparable
    gradient by gradient) — to float32 rounding of a different association; free-running, the estimates agree."""
    from aesmc_amd import _kernels, losses
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    B, K = 8, 4096

    def run(wide, T, forced=None):
        model = LgssmNd(128, dtype=torch.float32, affine=True, validate_args=False, emission_scale=0.05).tune_proposal().to(hip_device)
        observations = model.simulate(T, B, seed=2)
        calls = {"wide": 0, "backward": 0}
        recorded = []
        real, real_bwd, real_step = provider.affine_propagate_wide, provider.affine_step_backward_wide, provider.resample_step
        if wide:
            def counting(*args, **kwargs):
                out = real(*args, **kwargs)
                calls["wide"] += out is not None
                return out

            def counting_bwd(*args, **kwargs):
                calls["backward"] += 1
                return real_bwd(*args, **kwargs)
            monkeypatch.setattr(provider, "affine_propagate_wide", counting)
            monkeypatch.setattr(provider, "affine_step_backward_wide", counting_bwd)
        else:
            monkeypatch.setattr(provider, "affine_wide_covers", lambda *a, **k: False)

        def resampling(*args, **kwargs):
            out = real_step(*args, **kwargs)
            if forced is not None:
                out[0].copy_(forced[len(recorded)])      # the other run's ancestors (this run's own log-sum-exp)
            recorded.append(out[0].clone())
            return out
        monkeypatch.setattr(provider, "resample_step", resampling)
        np.random.seed(4)
        torch.manual_seed(4)
        loss = losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
        loss.backward()
        monkeypatch.undo()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, calls, recorded
    T = 5
    wide_loss, wide_grads, calls, ancestors = run(True, T)
    assert calls == {"wide": T - 1, "backward": T - 1}, calls
    assert len(ancestors) == T - 1
    forced_loss, forced_grads, _, _ = run(False, T, forced=ancestors)
    assert abs(wide_loss - forced_loss) <= 1e-5 * abs(forced_loss)
    assert set(wide_grads) == set(forced_grads) and len(wide_grads) >= 4
    for name, g in forced_grads.items():
        scale = float(g.abs().max()) + 1e-30
        assert float((wide_grads[name] - g).abs().max()) <= 2e-3 * scale, (name, float((wide_grads[name] - g).abs().max()), scale)
    free_loss, free_grads, _, _ = run(False, T)
    assert abs(wide_loss - free_loss) <= 2e-3 * abs(free_loss)
    assert all(bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0 for g in wide_grads.values())


# ---- K19: the wide backward's element-wise parts (linear_gaussian_wide_backward.hip) against the PyTorch operations they replace ----
@pytest.mark.parametrize("shape", [(2, 512), (3, 4096), (1, 256)])
@pytest.mark.parametrize("with_base", [False, True])
def test_the_wide_adjoints_elementwise_launches_equal_the_pytorch_operations(kernels, hip_device, shape, with_base):
    """aesmc_wide_adjoint_scale / _merge — residual, adjoint in place, sum_j d_j^2, the rows' sums, the arriving gradient
    merged — against the same steps written with PyTorch operations (autograd of aesmc/state.py:114-155 between the
    products): the element-wise results bit for bit (the same float32 operations in the same order), the sums to
    float32 rounding of another association."""
    B, K = shape
    d = 128
    gen = torch.Generator(device=hip_device).manual_seed(B * 1000 + K)
    make = lambda *s: torch.randn(*s, device=hip_device, generator=gen)
    weight, scale = make(B, K).abs() * 1e-3, torch.tensor(0.7, device=hip_device)
    loc, base = make(B, K, d), make(B, d)
    u = loc.clone()
    sq, rows = kernels.wide_adjoint_scale(u, weight, scale, True, True, base=base if with_base else None)
    resid = (base.unsqueeze(1) - loc) if with_base else loc
    want_u = resid * (weight / (scale * scale)).unsqueeze(2)
    assert torch.equal(u, want_u)
    assert torch.allclose(sq, resid.square().sum(2), rtol=2e-6, atol=0)
    assert torch.allclose(rows, want_u.sum(1), rtol=1e-5, atol=1e-6 * float(want_u.abs().max()) * (K ** 0.5))
    # the merge: location, value, offsets' rows, what later steps sent
    loc_p, value, arriving, later = make(B, K, d), make(B, K, d), make(B, K, d), make(B, K, d)
    u_p, at_x = loc_p.clone(), arriving.clone()
    sq_p, rows_p, rows_x = kernels.wide_adjoint_merge(u_p, at_x, weight, scale, True, True, True, value=value,
                                                      base=base if with_base else None, add=later if with_base else None)
    resid = (value - base.unsqueeze(1)) - loc_p if with_base else value - loc_p
    want_p = resid * (weight / (scale * scale)).unsqueeze(2)
    want_x = ((later + arriving) if with_base else arriving) - want_p
    assert torch.equal(u_p, want_p)
    assert torch.equal(at_x, want_x)
    assert torch.allclose(sq_p, resid.square().sum(2), rtol=2e-6, atol=0)
    bound = lambda t: 1e-6 * float(t.abs().max()) * (K ** 0.5)
    assert torch.allclose(rows_p, want_p.sum(1), rtol=1e-5, atol=bound(want_p))
    assert torch.allclose(rows_x, want_x.sum(1), rtol=1e-5, atol=bound(want_x))
    # a residual already formed (no value): the kernel takes u_p as it comes
    u_q, at_q = loc_p.clone(), arriving.clone()
    kernels.wide_adjoint_merge(u_q, at_q, weight, scale, False, False, False)
    assert torch.equal(u_q, loc_p * (weight / (scale * scale)).unsqueeze(2))
    assert torch.equal(at_q, arriving - u_q)
    # rows that are not whole tiles: declined by the wrapper's test, the caller keeps its PyTorch operations
    assert not kernels._wide_adjoint_covers(make(2, 384, d), make(2, 384), scale)


@pytest.mark.parametrize("dim,K", [(24, 1000), (64, 512), (192, 320)])
def test_models_of_other_widths_run_and_train_through_the_matrix_core_step(hip_device, monkeypatch, dim, K):
    """An LGSSM with rows of 24 / 64 / 192 values (AffineNormal callables), forward and under autograd: every resampled
    timestep is K17g + K18g, its backward the recomputing `affine_step_backward_wide` (library products, PyTorch
    element-wise parts at these widths); the loss and every parameter gradient equal the GEMM route's — the same model
    with the matrix-core step switched off, handed this run's ancestors (teacher forcing) — to what float32 log-weights of
    order 1e2 .. 1e3 allow two different summation orders to agree on after the softmax (2 % of a gradient's largest entry;
    the backward itself is held to float64 autograd at 5e-4 in test_the_wide_steps_backward_equals_float64_autograd).
    K = 1000 is not a multiple of 32 (masked tail)."""
    from aesmc_amd import _kernels, losses
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    B, T = 4, 4

    def run(wide, forced=None):
        model = LgssmNd(dim, dtype=torch.float32, affine=True, validate_args=False, emission_scale=0.5).tune_proposal().to(hip_device)
        observations = model.simulate(T, B, seed=2)
        calls = {"wide": 0, "backward": 0}
        recorded = []
        real, real_bwd, real_step = provider.affine_propagate_wide, provider.affine_step_backward_wide, provider.resample_step
        if wide:
            def counting(*args, **kwargs):
                out = real(*args, **kwargs)
                calls["wide"] += out is not None
                return out

            def counting_bwd(*args, **kwargs):
                calls["backward"] += 1
                return real_bwd(*args, **kwargs)
            monkeypatch.setattr(provider, "affine_propagate_wide", counting)
            monkeypatch.setattr(provider, "affine_step_backward_wide", counting_bwd)
        else:
            monkeypatch.setattr(provider, "affine_wide_covers", lambda *a, **k: False)

        def resampling(*args, **kwargs):
            out = real_step(*args, **kwargs)
            if forced is not None:
                out[0].copy_(forced[len(recorded)])
            recorded.append(out[0].clone())
            return out
        monkeypatch.setattr(provider, "resample_step", resampling)
        np.random.seed(4)
        torch.manual_seed(4)
        loss = losses.get_loss(observations, K, "aesmc", model.initial, model.transition, model.emission, model.proposal)
        loss.backward()
        monkeypatch.undo()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, calls, recorded

    wide_loss, wide_grads, calls, ancestors = run(True)
    assert calls == {"wide": T - 1, "backward": T - 1}, calls
    forced_loss, forced_grads, _, _ = run(False, forced=ancestors)
    assert abs(wide_loss - forced_loss) <= 2e-5 * abs(forced_loss), (wide_loss, forced_loss)
    assert set(wide_grads) == set(forced_grads) and len(wide_grads) >= 4
    for name, g in forced_grads.items():
        scale = float(g.abs().max()) + 1e-30
        assert float((wide_grads[name] - g).abs().max()) <= 2e-2 * scale, (name, float((wide_grads[name] - g).abs().max()), scale)


@pytest.mark.parametrize("dim", [24, 50])
def test_a_wide_model_written_with_matmuls_reaches_the_matrix_core_step_unedited(hip_device, monkeypatch, dim):
    """The reference's own style at a width the item kernels do not take — `Normal(x @ W.t() + c, s)` callables, no
    library class (test/models/lgssm.py:40's pattern, d = 24 / 50) — recorded on the lazy latents like the small maps and
    weighed by K17g + K18g: every resampled timestep is the matrix-core launch, and the estimate equals the eager PyTorch
    evaluation of the same model (lazy latents off) to float32 rounding of another summation order."""
    from aesmc_amd import _kernels, inference
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    model = LgssmNd(dim, dtype=torch.float32, affine=False, validate_args=False, emission_scale=0.5).tune_proposal().to(hip_device)
    B, K, T = 3, 200, 5
    observations = model.simulate(T, B, seed=2)
    calls = {"wide": 0}
    real = provider.affine_propagate_wide

    def counting(*args, **kwargs):
        out = real(*args, **kwargs)
        calls["wide"] += out is not None
        return out
    monkeypatch.setattr(provider, "affine_propagate_wide", counting)

    def run():
        np.random.seed(4)
        torch.manual_seed(4)
        with torch.no_grad():
            return inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal, K,
                                   return_log_marginal_likelihood=True, return_latents=False, return_original_latents=True)
    fused = run()
    assert calls["wide"] == T - 1, calls
    with inference.lazy_gather(False):
        plain = run()
    assert calls["wide"] == T - 1
    for t in range(2):      # (identical weights at the first resampling; later steps only while no ancestor flips)
        a, b = fused["original_latents"][t], plain["original_latents"][t]
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), t
    lml_f, lml_p = fused["log_marginal_likelihood"], plain["log_marginal_likelihood"]
    assert float(((lml_f - lml_p).abs() / lml_p.abs().clamp_min(1.0)).max()) <= 2e-3


@pytest.mark.parametrize("widths", [(64, 64), (200, 72), (256, 256)])
def test_the_matrix_core_step_at_a_million_particles_against_float64(kernels, hip_device, widths):
    """K17g + K18g at B K = 2^20 particles (B = 64, K = 16384: configs[4]'s particle count) — sizes the C oracle does not
    finish in seconds — through properties that do not depend on the size: x_t - eps s_q is the proposal's location and the
    log-weight is the three Normal log-densities, both evaluated in float64 by PyTorch on the gathered rows; every row of
    x_t written exactly once (no NaN left of the poison fill)."""
    dx, dy = widths
    B, K = 64, 16384
    gen = torch.Generator(device=hip_device).manual_seed(dx + dy)
    make = lambda *s: torch.randn(*s, device=hip_device, generator=gen)
    x_prev, eps, y, off_q, off_g = make(B, K, dx), make(B, K, dx), make(B, dy), make(B, dx), make(dy)
    eye = torch.eye(dx, device=hip_device)
    A = 0.9 * eye + 0.3 / dx ** 0.5 * make(dx, dx)
    Q = 0.45 * eye + 0.3 / dx ** 0.5 * make(dx, dx)
    C = make(dy, dx) / dx ** 0.5
    scales = tuple(torch.tensor(v, device=hip_device) for v in (1.0, 0.5, 0.7))
    anc = _ancestors(B, K, hip_device, seed=5, spread=1.0)
    out_x = torch.full((B, K, dx), float("nan"), device=hip_device)
    lw = kernels.affine_propagate_wide(x_prev, eps, y, (A, None), (C, off_g), (Q, off_q), scales, out_x, ancestors=anc)
    assert lw is not None and bool(torch.isfinite(out_x).all()) and bool(torch.isfinite(lw).all())
    rows = slice(0, B, 9)      # a sample of batch rows in float64 (the full tensors would be 3 x 2 GB)
    moved = torch.gather(x_prev[rows], 1, anc[rows].unsqueeze(-1).expand(-1, -1, dx)).double()
    loc_q = moved @ Q.double().t() + off_q[rows].double().unsqueeze(1)
    want_x = loc_q + eps[rows].double() * 0.7
    assert float((out_x[rows].double() - want_x).abs().max()) <= 2e-5 * float(want_x.abs().max())
    x64 = out_x[rows].double()
    logn = lambda v, loc, s: torch.distributions.Normal(loc, s).log_prob(v).sum(-1)
    want_lw = logn(x64, moved @ A.double().t(), 1.0) + logn(y[rows].double().unsqueeze(1), x64 @ C.double().t() + off_g.double(), 0.5) \
        - logn(x64, loc_q, 0.7)
    scale = want_lw.abs().clamp_min(1.0)
    assert float(((lw[rows].double() - want_lw).abs() / scale).max()) <= 2e-5
    assert kernels.read_flags(hip_device) == 0
