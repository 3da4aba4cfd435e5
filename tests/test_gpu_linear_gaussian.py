"""GPU parity tests of the linear-Gaussian propagation kernels (K8 particle_affine, K9
affine_normal_rsample, K10 affine_normal_logweight), through the C ABI:

  * K8 and K9 against oracle/smc_core.c BIT FOR BIT (the location is one fma chain in a fixed order,
    the draw a rounded product plus a sum: libm's fmaf / fma restate both exactly);
  * K10 against the C oracle within the rounding of the device's logf (everything else in it is
    exactly rounded IEEE arithmetic in a fixed order) and within float rounding of the unfused
    route on the device (K8 x 3 + K5, which divides per element where K10 divides once per term);
  * K9 bit for bit against K8 + K6.
Shapes cover ragged tails (particles not a multiple of the 256- / 512-particle tile), one particle
per row, every padded extent class (4, 8, 12, 16), unequal latent / observation extents, transposed
weight views, per-row, shared and absent offsets.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu

SHAPES = [(3, 700, 10, 10), (2, 513, 5, 3), (5, 64, 16, 16), (4, 1, 1, 1), (1, 1000, 3, 7), (7, 300, 12, 2),
          (2, 2048, 8, 8), (300, 5, 4, 4), (1, 1, 2, 16)]


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


def operands(B, K, dx, dy, dtype, device, seed):
    rng = np.random.RandomState(seed)
    r = lambda *shape: rng.randn(*shape).astype(dtype)
    host = {"x_prev": r(B, K, dx), "x": r(B, K, dx), "eps": r(B, K, dx), "y": r(B, dy),
            "A": (0.9 * np.eye(dx) + 0.1 * rng.randn(dx, dx)).astype(dtype),
            "Q": (0.45 * np.eye(dx) + 0.1 * rng.randn(dx, dx)).astype(dtype), "C": (0.3 * rng.randn(dy, dx)).astype(dtype),
            "off_q": r(B, dx), "off_g": r(dy), "s_p": np.asarray(1.0, dtype), "s_g": np.asarray(0.5, dtype),
            "s_q": np.asarray(0.7, dtype)}
    return host, {key: torch.from_numpy(np.ascontiguousarray(value)).to(device) for key, value in host.items()}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_particle_affine_equals_the_c_oracle_bit_for_bit(kernels, hip_device, dtype, shape):
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=B * K + dx)
    loc = kernels.particle_affine(o["x_prev"], o["Q"], o["off_q"])
    np.testing.assert_array_equal(loc.cpu().numpy(), c_oracle.particle_affine(n["x_prev"], n["Q"], offset=n["off_q"]))
    shared = kernels.particle_affine(o["x"], o["C"], o["off_g"])          # [dy] offset shared by all rows
    np.testing.assert_array_equal(shared.cpu().numpy(), c_oracle.particle_affine(n["x"], n["C"], offset=n["off_g"]))
    # two inputs through TRANSPOSED weight views on top of a base: the adjoint of the step's two maps
    two = kernels.particle_affine(o["x_prev"], o["A"].t(), None, o["x"], o["Q"].t(), base=o["eps"])
    want = c_oracle.particle_affine(n["x_prev"], n["A"].T.copy(), x2=n["x"], w2=n["Q"].T.copy(), base=n["eps"])
    np.testing.assert_array_equal(two.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_affine_rsample_equals_the_c_oracle_and_the_unfused_route(kernels, hip_device, dtype, shape):
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=B + K + dx)
    draw = kernels.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"])
    want = c_oracle.affine_rsample(n["x_prev"], n["Q"], n["off_q"], n["eps"], float(n["s_q"]))
    np.testing.assert_array_equal(draw.cpu().numpy(), want)
    loc = kernels.particle_affine(o["x_prev"], o["Q"], o["off_q"])
    assert torch.equal(draw, kernels.normal_rsample(o["eps"], loc, o["s_q"].expand_as(loc)))
    bare = kernels.affine_rsample(o["x_prev"], o["Q"], None, o["eps"], o["s_q"])
    np.testing.assert_array_equal(bare.cpu().numpy(),
                                  c_oracle.affine_rsample(n["x_prev"], n["Q"], None, n["eps"], float(n["s_q"])))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_affine_logweight_matches_the_c_oracle_and_equals_the_unfused_route(kernels, hip_device, dtype, shape):
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=B * dx + K)
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    lw = kernels.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
    want = c_oracle.affine_logweight(n["x_prev"], n["x"], n["y"], (n["A"], None), (n["C"], n["off_g"]),
                                     (n["Q"], n["off_q"]), float(n["s_p"]), float(n["s_g"]), float(n["s_q"]))
    rtol = 5e-7 if dtype == np.float32 else 1e-15   # the device's log(sigma) against glibc's, times d
    np.testing.assert_allclose(lw.cpu().numpy(), want, rtol=rtol, atol=rtol * max(1.0, float(np.abs(want).max())))
    loc_p = kernels.particle_affine(o["x_prev"], o["A"])
    loc_g = kernels.particle_affine(o["x"], o["C"], o["off_g"])
    loc_q = kernels.particle_affine(o["x_prev"], o["Q"], o["off_q"])
    y_expanded = o["y"].unsqueeze(1).expand(B, K, dy)
    route = kernels.normal_logweight(o["x"], loc_p, o["s_p"].expand_as(loc_p), y_expanded, loc_g,
                                     o["s_g"].expand_as(loc_g), loc_q, o["s_q"].expand_as(loc_q))
    assert route is not None
    rtol = 2e-6 if dtype == np.float32 else 1e-14
    scale = max(1.0, float(route.abs().max()))
    assert float((lw - route).abs().max()) <= rtol * scale


def test_full_size_affine_kernels_agree_with_library_matmul(kernels, hip_device):
    """BASELINE configs[3]'s shape (B=1024, K=4096, d=10): the fused kernels against the float64
    library route (size-independent property: a location is linear in its inputs)."""
    B, K, d = 1024, 4096, 10
    gen = torch.Generator(device=hip_device).manual_seed(3)
    x_prev = torch.randn(B, K, d, device=hip_device, generator=gen)
    eps = torch.randn(B, K, d, device=hip_device, generator=gen)
    Q = 0.45 * torch.eye(d, device=hip_device) + 0.05 * torch.randn(d, d, device=hip_device, generator=gen)
    off = torch.randn(B, d, device=hip_device, generator=gen)
    scale = torch.tensor(0.7, device=hip_device)
    draw = kernels.affine_rsample(x_prev, Q, off, eps, scale)
    want = (x_prev.double() @ Q.double().t() + off.double().unsqueeze(1)) + eps.double() * 0.7
    assert float((draw.double() - want).abs().max()) < 5e-6
    loc = kernels.particle_affine(x_prev, Q, off)
    assert torch.equal(draw, kernels.normal_rsample(eps, loc, scale.expand_as(loc)))
    doubled = kernels.particle_affine(2 * x_prev, Q, 2 * off)     # exact: scaling by two commutes with rounding
    assert torch.equal(doubled, 2 * loc)


# ---- end to end: a model whose callables return AffineNormal against the CPU port of the reference ----
from aesmc_amd import inference, losses  # noqa: E402
from aesmc_amd.testing import models, replay  # noqa: E402


def _parts(model):
    return model.initial, model.transition, model.emission, model.proposal


@pytest.mark.parametrize("B,K,T,d", [(3, 300, 4, 5), (2, 1024, 6, 10), (5, 64, 3, 16), (4, 17, 5, 1), (2, 600, 3, 3)])
def test_affine_callables_match_the_cpu_port_draw_for_draw(hip_device, B, K, T, d):
    """The CPU port of the reference runs the LGSSM with PLAIN Normal(matmul) callables and records its
    draws; the product replays them on a model whose callables return AffineNormal (kernels K9 / K10,
    K8 at time 0).  float64: ancestor indices exact, per-step log-weights and log Z to 1e-10."""
    from oracle import reference_port
    dtype = torch.float64
    cpu_model = models.LgssmNd(d, seed=0, dtype=dtype, state=reference_port).tune_proposal()
    observations = cpu_model.simulate(T, B, seed=1)
    np.random.seed(7)
    torch.manual_seed(7)
    flags = dict(return_log_marginal_likelihood=True, return_log_weights=True, return_ancestral_indices=True,
                 return_original_latents=True)
    with replay.record() as tape:
        want = reference_port.infer("smc", observations, *_parts(cpu_model), K, **flags)
    model = models.LgssmNd(d, seed=0, dtype=dtype, affine=True).to(hip_device).tune_proposal()
    launches = _count_affine_launches()
    with replay.replay(tape), launches:
        got = inference.infer("smc", [o.to(hip_device) for o in observations], *_parts(model), K, **flags)
    assert launches.count["affine_rsample"] == T - 1 and launches.count["affine_logweight"] == T - 1
    for a, b in zip(got["ancestral_indices"], want["ancestral_indices"]):
        assert torch.equal(a.cpu(), b)
    for a, b in zip(got["log_weights"], want["log_weights"]):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-10, atol=1e-10)
    for a, b in zip(got["latents"], want["latents"]):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(got["log_marginal_likelihood"].cpu(), want["log_marginal_likelihood"],
                               rtol=1e-10, atol=1e-10)


class _count_affine_launches:
    """Counts the provider's K9 / K10 launches inside the context (the fused route must be the one taken)."""

    def __init__(self):
        self.count = {"affine_rsample": 0, "affine_logweight": 0}

    def __enter__(self):
        from aesmc_amd import _kernels
        self.provider = _kernels.get()
        self.saved = {name: getattr(self.provider, name) for name in self.count}
        for name in self.count:
            def spy(*args, _name=name, **kwargs):
                self.count[_name] += 1
                return self.saved[_name](*args, **kwargs)
            setattr(self.provider, name, spy)
        return self

    def __exit__(self, *exc):
        for name in self.count:
            delattr(self.provider, name)
        return False


@pytest.mark.parametrize("algorithm,B,K,T,d", [("aesmc", 3, 300, 4, 5), ("aesmc", 2, 512, 5, 10), ("iwae", 3, 64, 3, 4)])
def test_affine_callables_loss_and_gradients_match_the_cpu_port(hip_device, algorithm, B, K, T, d):
    """get_loss + backward: the CPU port (plain callables, PyTorch autograd on the host) records its
    draws, the GPU replays them through AffineNormal callables.  float64: loss to 1e-10, every
    parameter gradient to 1e-8 of its largest entry."""
    from oracle import reference_port
    dtype = torch.float64
    cpu_model = models.LgssmNd(d, seed=0, dtype=dtype, state=reference_port).tune_proposal()
    observations = cpu_model.simulate(T, B, seed=1)
    np.random.seed(11)
    torch.manual_seed(11)
    with replay.record() as tape:
        want = reference_port.get_loss(observations, K, algorithm, *_parts(cpu_model))
    want.backward()
    model = models.LgssmNd(d, seed=0, dtype=dtype, affine=True).to(hip_device).tune_proposal()
    with replay.replay(tape):
        got = losses.get_loss([o.to(hip_device) for o in observations], K, algorithm, *_parts(model))
    got.backward()
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


def test_affine_normal_is_the_normal_it_stands_for(hip_device):
    """loc / scale / log_prob / rsample / expand of an AffineNormal equal those of
    Normal(source @ weight.T + offset, scale) built from the materialised location."""
    from aesmc_amd.linear_gaussian import AffineNormal
    gen = torch.Generator(device=hip_device).manual_seed(5)
    source = torch.randn(4, 33, 6, device=hip_device, dtype=torch.float64, generator=gen)
    weight = torch.randn(5, 6, device=hip_device, dtype=torch.float64, generator=gen)
    offset = torch.randn(4, 5, device=hip_device, dtype=torch.float64, generator=gen)
    scale = torch.tensor(0.3, device=hip_device, dtype=torch.float64)
    dist = AffineNormal(source, weight, scale, offset=offset)
    plain = torch.distributions.Normal(source @ weight.t() + offset.unsqueeze(1), scale)
    assert dist.batch_shape == plain.batch_shape and dist.event_shape == plain.event_shape
    torch.testing.assert_close(dist.loc, plain.loc, rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(dist.mean, plain.mean, rtol=1e-13, atol=1e-13)
    assert torch.equal(dist.scale, plain.scale.expand(plain.batch_shape))
    value = torch.randn(4, 33, 5, device=hip_device, dtype=torch.float64, generator=gen)
    torch.testing.assert_close(dist.log_prob(value), plain.log_prob(value), rtol=1e-12, atol=1e-12)
    torch.manual_seed(3)
    a = dist.rsample()
    torch.manual_seed(3)
    b = plain.rsample()
    torch.testing.assert_close(a, b, rtol=1e-13, atol=1e-13)
    assert dist.expand((2, 4, 33, 5)).batch_shape == (2, 4, 33, 5)
    wide = AffineNormal(torch.randn(2, 8, 40, device=hip_device), torch.randn(24, 40, device=hip_device), 1.0)
    assert wide.loc.shape == (2, 8, 24)     # beyond 16 x 16: the library's matmul


def test_affine_callables_float32_agree_with_matmul_callables(hip_device):
    """float32 at a configs[1]-like shape: the two ways of stating the model give the same ELBO to
    float32 accuracy and ancestor indices that differ only where a CDF comparison sits within
    rounding noise of flipping (the locations differ in their last bits: fma chain against the
    library's matmul)."""
    B, K, T, d = 16, 1024, 12, 10
    results = {}
    for affine in (False, True):
        model = models.LgssmNd(d, seed=0, affine=affine).to(hip_device).tune_proposal()
        observations = [o.to(hip_device) for o in model.simulate(T, B, seed=1)]
        np.random.seed(2)
        torch.manual_seed(2)
        with inference.lazy_gather(affine):      # the matmul statement evaluated by PyTorch itself
            results[affine] = inference.infer("smc", observations, *_parts(model), K,
                                              return_log_marginal_likelihood=True, return_ancestral_indices=True,
                                              return_latents=False)
    a, b = results[False], results[True]
    first = a["ancestral_indices"][0], b["ancestral_indices"][0]
    agree = float((first[0] == first[1]).double().mean())
    assert agree > 0.995 and int((first[0] - first[1]).abs().max()) <= 1
    za, zb = a["log_marginal_likelihood"], b["log_marginal_likelihood"]
    assert float(((za - zb).abs() / za.abs()).max().detach()) < 5e-3     # a flipped index moves a row's later weights


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES + [(64, 4096, 10, 10)])
def test_particle_affine_backward_matches_float64_matmuls(kernels, hip_device, dtype, shape):
    """K11: grad_x = grad @ W is the K8 chain on the transposed weight (bit for bit the C oracle's);
    grad_W = grad^T x summed over all particles on the matrix cores, against the float64 contraction
    (float32: the fixed-order partial sums are within 1e-5 of the gradient's scale), and identical from
    run to run."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=B + 3 * K)
    grad = torch.from_numpy(np.random.RandomState(1).randn(B, K, dy).astype(dtype)).to(hip_device)
    gx, gw, goff = kernels.particle_affine_backward(grad, o["x"], o["C"], need_offset=True)
    want_off = grad.double().sum(dim=1)
    assert float((goff.double() - want_off).abs().max()) <= (1e-5 if dtype == np.float32 else 1e-13) * \
        max(1.0, float(want_off.abs().max()))
    assert torch.equal(goff, kernels.particle_affine_backward(grad, o["x"], o["C"], False, False, True)[2])
    want_gx = c_oracle.particle_affine(grad.cpu().numpy(), n["C"].T.copy())
    np.testing.assert_array_equal(gx.cpu().numpy(), want_gx)
    want_gw = grad.double().reshape(-1, dy).t() @ o["x"].double().reshape(-1, dx)
    scale = max(1.0, float(want_gw.abs().max()))
    tolerance = 1e-5 if dtype == np.float32 else 1e-13
    assert float((gw.double() - want_gw).abs().max()) <= tolerance * scale
    again = kernels.particle_affine_backward(grad, o["x"], o["C"], need_x=False)[1]
    assert torch.equal(gw, again)
    assert torch.equal(kernels.outer_sum(grad, o["x"]), gw)


def test_particle_affine_operator_gradients(hip_device):
    """autograd through aesmc_amd.linear_gaussian.particle_affine (K8 forward, K11 backward) against
    PyTorch's own autograd over matmul, float64."""
    from aesmc_amd.linear_gaussian import particle_affine
    gen = torch.Generator(device=hip_device).manual_seed(9)
    make = lambda *shape: torch.randn(*shape, device=hip_device, dtype=torch.float64, generator=gen).requires_grad_(True)
    x, weight, offset = make(3, 257, 7), make(5, 7), make(3, 5)
    out = particle_affine(x, weight, offset)
    upstream = torch.randn(out.shape, device=hip_device, dtype=torch.float64, generator=gen)
    got = torch.autograd.grad(out, (x, weight, offset), upstream)
    ref = torch.autograd.grad(x @ weight.t() + offset.unsqueeze(1), (x, weight, offset), upstream)
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("via_lse", [False, True])
@pytest.mark.parametrize("shape", SHAPES + [(16, 4096, 10, 10)])
def test_affine_logweight_backward_matches_autograd_and_the_unfused_route(kernels, hip_device, dtype, via_lse, shape):
    """K12 against (i) PyTorch's autograd over the float64 expression (every gradient: both latents,
    the observation, three weights, three offsets, three scales) and (ii) the unfused device route
    (K8 x 3, K5's backward, K8 transposed, K11).  With `via_lse` the incoming gradient is K1's softmax
    term formed inside the kernel from (lw, lse, grad_lse)."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=2 * B + K + dy)
    off_p = torch.from_numpy(np.random.RandomState(4).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    rng = np.random.RandomState(8)
    need = [True] * 12
    if via_lse:
        lw = kernels.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
        _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
        grad_lse = torch.from_numpy(rng.randn(B).astype(dtype)).to(hip_device)
        incoming = dict(lw=lw, lse=lse, grad_lse=grad_lse)
        g = grad_lse.double().unsqueeze(1) * torch.exp(lw.double() - lse.double().unsqueeze(1))
    else:
        grad_lw = torch.from_numpy(rng.randn(B, K).astype(dtype)).to(hip_device)
        incoming = dict(grad_lw=grad_lw)
        g = grad_lw.double()
    got = kernels.affine_logweight_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, **incoming)
    route = kernels.affine_logweight_backward_unfused(o["x_prev"], o["x"], o["y"], *terms, scales, need, **incoming)
    # float64 autograd over the expression itself
    leaves = [t.detach().double().requires_grad_(True) for t in
              (o["x_prev"], o["x"], o["y"], o["A"], off_p, o["C"], o["off_g"], o["Q"], o["off_q"], *scales)]
    xp, xx, yy, A, op, C, og, Q, oq, sp, sg, sq = leaves
    normal = torch.distributions.Normal
    value = (normal(xp @ A.t() + op, sp).log_prob(xx).sum(-1) +
             normal(xx @ C.t() + og, sg).log_prob(yy.unsqueeze(1)).sum(-1) -
             normal(xp @ Q.t() + oq.unsqueeze(1), sq).log_prob(xx).sum(-1))
    want = torch.autograd.grad(value, leaves, grad_outputs=g)
    names = ("x_prev", "x", "y", "A", "off_p", "C", "off_g", "Q", "off_q", "s_p", "s_g", "s_q")
    tolerance = 2e-5 if dtype == np.float32 else 1e-11
    for name, a, b, c in zip(names, got, want, route):
        assert a is not None and a.shape == b.shape, name
        scale = max(1.0, float(b.abs().max()))
        assert float((a.double() - b).abs().max()) <= tolerance * scale, (name, "vs autograd")
        assert float((a.double() - c.double()).abs().max()) <= tolerance * scale, (name, "vs unfused route")
    again = kernels.affine_logweight_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, **incoming)
    for a, b in zip(got, again):
        assert torch.equal(a, b)     # fixed summation order: reproducible
    # only some gradients wanted: the others are not produced
    some = [False] * 12
    some[1] = some[7] = True
    partial = kernels.affine_logweight_backward(o["x_prev"], o["x"], o["y"], *terms, scales, some, **incoming)
    assert [t is not None for t in partial] == some
    assert torch.equal(partial[1], got[1]) and torch.equal(partial[7], got[7])


# ---- K15: the draw and its log-weight in one launch -----------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES + [(16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (40, 30, 6, 9)])
def test_affine_propagate_equals_the_draw_kernel_then_the_weight_kernel_bit_for_bit(kernels, hip_device, dtype, shape):
    """K15 against K9 followed by K10 on the device — the same chains, so the same bits, for the draw and
    for the log-weight (one and two particles per lane, table and per-lane row paths, ragged tails) — and,
    through them, against the C oracle; the draw may be written over the noise's own buffer."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=5 * B + K + dy)
    off_p = torch.from_numpy(np.random.RandomState(6).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    draw = kernels.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"])
    lw = kernels.affine_logweight(o["x_prev"], draw, o["y"], *terms, scales)
    out_x = torch.full_like(draw, float("nan"))
    got = kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=out_x)
    assert torch.equal(out_x, draw)
    assert torch.equal(got, lw)
    want = c_oracle.affine_rsample(n["x_prev"], n["Q"], n["off_q"], n["eps"], float(n["s_q"]))
    assert np.array_equal(out_x.cpu().numpy(), want)
    noise = o["eps"].clone()
    again = kernels.affine_propagate(o["x_prev"], noise, o["y"], *terms, scales, out_x=noise)     # in place
    assert torch.equal(noise, draw) and torch.equal(again, lw)
    with pytest.raises(ValueError):
        kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=o["x_prev"])
    with pytest.raises(ValueError):
        kernels.affine_propagate(o["x_prev"], o["eps"], o["y"], *terms, scales, out_x=out_x[:, :, :-1])


@pytest.mark.parametrize("grad", [False, True])
def test_a_deferred_draw_gives_the_very_same_run_on_the_device(hip_device, grad):
    """defer_draw on the proposal (K15 draws and weighs) against the immediate draw (K9, then K10): every
    number of the run identical — latents, ancestors, evidence, gradients, RNG consumption."""
    from aesmc_amd import inference
    from aesmc_amd.testing.models import LgssmNd
    runs = {}
    for defer in (False, True):
        model = LgssmNd(10, dtype=torch.float32, affine=True, defer_draw=defer).tune_proposal().to(hip_device)
        observations = model.simulate(5, 4, seed=3)
        torch.manual_seed(11)
        np.random.seed(11)
        with torch.set_grad_enabled(grad):
            out = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal,
                                  1300, return_log_marginal_likelihood=True, return_latents=True,
                                  return_log_weight=not grad, return_ancestral_indices=True)
        if grad:
            (-out["log_marginal_likelihood"].mean()).backward()
        after = (torch.rand(1, device=hip_device).item(), np.random.uniform())
        runs[defer] = (out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, after)
    (a, grads_a, rng_a), (b, grads_b, rng_b) = runs[False], runs[True]
    assert rng_a == rng_b
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    for x, y in zip(a["latents"] + a["ancestral_indices"], b["latents"] + b["ancestral_indices"]):
        assert torch.equal(x, y)
    assert sorted(grads_a) == sorted(grads_b) and (not grad or grads_a)
    for name in grads_a:
        assert torch.equal(grads_a[name], grads_b[name]), name


# ---- K14: the whole backward of a step whose latent is the proposal's draw ----------------------------------
def _step_reference(o, off_p, x, g, grad_x):
    """float64 autograd over the step with x rebuilt as the draw loc_q(x_prev) + s_q eps (eps held fixed)."""
    leaves = [t.detach().double().requires_grad_(True) for t in
              (o["x_prev"], o["y"], o["A"], off_p, o["C"], o["off_g"], o["Q"], o["off_q"], o["s_p"], o["s_g"], o["s_q"])]
    xp, yy, A, op, C, og, Q, oq, sp, sg, sq = leaves
    loc_q = xp @ Q.t() + oq.unsqueeze(1)
    with torch.no_grad():
        eps = (x.double() - loc_q) / sq
    draw = loc_q + sq * eps
    normal = torch.distributions.Normal
    value = (normal(xp @ A.t() + op, sp).log_prob(draw).sum(-1) +
             normal(draw @ C.t() + og, sg).log_prob(yy.unsqueeze(1)).sum(-1) -
             normal(loc_q, sq).log_prob(draw).sum(-1))
    total = (value * g).sum()
    if grad_x is not None:
        total = total + (draw * grad_x.double()).sum()
    grads = torch.autograd.grad(total, leaves)
    slots = (0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)
    out = [None] * 12
    for slot, grad in zip(slots, grads):
        out[slot] = grad
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("with_grad_x", [False, True])
@pytest.mark.parametrize("shape", SHAPES + [(16, 4096, 10, 10), (300, 4096, 10, 10), (520, 2100, 8, 8), (2, 2048, 12, 12)])
def test_step_backward_matches_autograd_and_the_launches_it_replaces(kernels, hip_device, dtype, with_grad_x, shape):
    """K14 against (i) PyTorch's float64 autograd over the step with x_t rebuilt as the proposal's draw and
    (ii) the launches it replaces (K12 with x_t's gradient, the accumulation, K11 through the draw):
    every gradient — x_{t-1}, observation, three weights, three offsets, three scales — and None for x_t.
    The last shapes reach the two-particles-per-lane launch (2^20 particles and more) with exact extents:
    offsets' gradients out of the matrix cores' spare column where a tile lies inside one batch row
    (K = 4096) and by the pass over the tile where it does not (K = 2100)."""
    B, K, dx, dy = shape
    n, o = operands(B, K, dx, dy, dtype, hip_device, seed=3 * B + K + dx)
    off_p = torch.from_numpy(np.random.RandomState(4).randn(dx).astype(dtype)).to(hip_device)
    terms = ((o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    x = kernels.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"])      # the draw (K9)
    lw = kernels.affine_logweight(o["x_prev"], x, o["y"], *terms, scales)
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    rng = np.random.RandomState(9)
    grad_lse = torch.from_numpy(rng.randn(B).astype(dtype)).to(hip_device)
    grad_x = torch.from_numpy(rng.randn(B, K, dx).astype(dtype)).to(hip_device) if with_grad_x else None
    need = [True] * 12
    need[1] = False
    got = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                       grad_x=grad_x)
    route = kernels.affine_step_backward_unfused(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse,
                                                 grad_lse=grad_lse, grad_x=grad_x)
    g = grad_lse.double().unsqueeze(1) * torch.exp(lw.double() - lse.double().unsqueeze(1))
    want = _step_reference(o, off_p, x, g, grad_x)
    names = ("x_prev", "x", "y", "A", "off_p", "C", "off_g", "Q", "off_q", "s_p", "s_g", "s_q")
    tolerance = 3e-5 if dtype == np.float32 else 1e-11
    assert got[1] is None and route[1] is None
    for name, a, b, c in zip(names, got, want, route):
        if name == "x":
            continue
        assert a is not None and a.shape == b.shape, name
        scale = max(1.0, float(b.abs().max()))
        assert float((a.double() - b).abs().max()) <= tolerance * scale, (name, "vs autograd")
        assert float((a.double() - c.double()).abs().max()) <= tolerance * scale, (name, "vs the launches it replaces")
    again = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_lse=grad_lse,
                                         grad_x=grad_x)
    for a, b in zip(got, again):
        assert (a is None and b is None) or torch.equal(a, b)     # fixed summation order: reproducible
    some = [False] * 12
    some[0] = some[7] = True            # no scale gradient: the kernel skips the proposal's location
    partial = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, some, lw, lse, grad_lse=grad_lse,
                                           grad_x=grad_x)
    assert [t is not None for t in partial] == some
    assert torch.equal(partial[0], got[0]) and torch.equal(partial[7], got[7])
    if with_grad_x:                     # only later steps' gradient arrives (the ELBO term absent)
        only = kernels.affine_step_backward(o["x_prev"], x, o["y"], *terms, scales, need, lw, lse, grad_x=grad_x)
        want_only = _step_reference(o, off_p, x, torch.zeros(B, K, dtype=torch.float64, device=hip_device), grad_x)
        for name, a, b in zip(names, only, want_only):
            if name != "x":
                assert float((a.double() - b).abs().max()) <= tolerance * max(1.0, float(b.abs().max())), name


def test_step_backward_validates_its_arguments(kernels, hip_device):
    n, o = operands(2, 300, 4, 3, np.float32, hip_device, seed=1)
    terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
    scales = (o["s_p"], o["s_g"], o["s_q"])
    lw = kernels.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    need = [True] * 12
    with pytest.raises(ValueError):      # x has no gradient slot
        kernels.affine_step_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, lw, lse,
                                     grad_lse=torch.ones(2, device=hip_device))
    need[1] = False
    with pytest.raises(ValueError):      # no gradient at all
        kernels.affine_step_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, lw, lse)
    with pytest.raises(ValueError):
        kernels.affine_step_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, lw, lse,
                                     grad_lse=torch.ones(3, device=hip_device))


def test_a_linear_gaussian_smc_step_is_one_backward_launch(hip_device):
    """get_loss + backward over AffineNormal callables: every step from the second on is K10 forward and ONE
    K14 launch backward — no K12, no backward of the draw — and x_t keeps reaching the callers as a tensor
    that requires grad (latents returned by `infer` still differentiate)."""
    from aesmc_amd import _kernels, inference, losses
    from aesmc_amd.testing.models import LgssmNd
    provider = _kernels.get()
    calls = {"affine_step_backward": 0, "affine_logweight_backward": 0, "particle_affine_backward": 0}
    originals = {name: getattr(provider, name) for name in calls}
    for name in calls:
        def spy(*args, _name=name, **kwargs):
            calls[_name] += 1
            return originals[_name](*args, **kwargs)
        setattr(provider, name, spy)
    try:
        T = 6
        model = LgssmNd(5, dtype=torch.float64, affine=True).tune_proposal().to(hip_device)
        observations = model.simulate(T, 3, seed=2)
        torch.manual_seed(3)
        np.random.seed(3)
        loss = losses.get_loss(observations, 300, "aesmc", model.initial, model.transition, model.emission,
                               model.proposal)
        loss.backward()
        assert calls["affine_step_backward"] == T - 1 and calls["affine_logweight_backward"] == 0
        assert calls["particle_affine_backward"] == 1       # time 0's emission location only
        torch.manual_seed(3)
        np.random.seed(3)
        result = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal,
                                 300, return_log_marginal_likelihood=True, return_latents=True, return_log_weight=False)
        assert all(latent.requires_grad for latent in result["latents"][1:])
        torch.testing.assert_close(-result["log_marginal_likelihood"].mean(), loss.detach(), rtol=1e-12, atol=1e-12)
    finally:
        for name, fn in originals.items():
            setattr(provider, name, fn)


@pytest.mark.parametrize("algorithm,B,K,T,d", [("aesmc", 3, 300, 4, 5), ("aesmc", 2, 512, 3, 10)])
def test_fused_nonlinear_model_matches_the_cpu_port_with_gradients(hip_device, algorithm, B, K, T, d):
    """BASELINE.json's nonlinear state-space model with its d x d maps through K8 (`fused=True`; the proposal net
    stays PyTorch's) against the CPU port running the plain PyTorch callables, draws replayed: float64 loss to
    1e-10, every parameter gradient to 1e-8 of its largest entry."""
    from oracle import reference_port
    dtype = torch.float64
    cpu_model = models.NonlinearSsm(d, hidden=24, seed=0, dtype=dtype, state=reference_port)
    observations = cpu_model.simulate(T, B, seed=1)
    np.random.seed(5)
    torch.manual_seed(5)
    with replay.record() as tape:
        want = reference_port.get_loss(observations, K, algorithm, *_parts(cpu_model))
    want.backward()
    model = models.NonlinearSsm(d, hidden=24, seed=0, dtype=dtype, fused=True).to(hip_device)
    with replay.replay(tape):
        got = losses.get_loss([o.to(hip_device) for o in observations], K, algorithm, *_parts(model))
    got.backward()
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-10, atol=1e-10)
    expected = dict(cpu_model.named_parameters())
    for name, parameter in model.named_parameters():
        reference = expected[name].grad
        if reference is None:
            continue
        scale = max(float(reference.abs().max()), 1e-30)
        assert float((parameter.grad.cpu() - reference).abs().max()) <= 1e-8 * scale, name


def test_random_shapes_through_every_linear_gaussian_kernel(kernels, hip_device):
    """60 random (B, K, dx, dy) — batch rows from 1 to 3000 particles, every extent 1..16, tiles ending
    anywhere, table and per-lane row paths, one and two particles per lane — through K8 / K9 (bitwise
    against the C oracle), K10 (tolerance), K11 / K12 (float64 autograd, incl. offset and scale gradients)."""
    rng = np.random.RandomState(2024)
    for case in range(60):
        B = int(rng.randint(1, 40))
        K = int(rng.choice([1, 2, 7, 33, 64, 85, 86, 100, 255, 256, 257, 511, 513, 1000, 3000]))
        if B * K > 60000:
            B = max(1, 60000 // K)
        dx, dy = int(rng.randint(1, 17)), int(rng.randint(1, 17))
        dtype = np.float64 if case % 3 == 0 else np.float32
        n, o = operands(B, K, dx, dy, dtype, hip_device, seed=1000 + case)
        tag = (case, B, K, dx, dy, dtype.__name__)
        draw = kernels.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"])
        want = c_oracle.affine_rsample(n["x_prev"], n["Q"], n["off_q"], n["eps"], float(n["s_q"]))
        assert np.array_equal(draw.cpu().numpy(), want), tag
        terms = ((o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]))
        scales = (o["s_p"], o["s_g"], o["s_q"])
        lw = kernels.affine_logweight(o["x_prev"], o["x"], o["y"], *terms, scales)
        want = c_oracle.affine_logweight(n["x_prev"], n["x"], n["y"], (n["A"], None), (n["C"], n["off_g"]),
                                         (n["Q"], n["off_q"]), float(n["s_p"]), float(n["s_g"]), float(n["s_q"]))
        rtol = 1e-6 if dtype == np.float32 else 1e-14
        assert float(np.abs(lw.cpu().numpy() - want).max()) <= rtol * max(1.0, float(np.abs(want).max())), tag
        grad_lw = torch.from_numpy(rng.randn(B, K).astype(dtype)).to(hip_device)
        need = [True] * 12
        need[4] = False       # no transition offset in `terms`
        got = kernels.affine_logweight_backward(o["x_prev"], o["x"], o["y"], *terms, scales, need, grad_lw=grad_lw)
        leaves = [t.detach().double().requires_grad_(True) for t in
                  (o["x_prev"], o["x"], o["y"], o["A"], o["C"], o["off_g"], o["Q"], o["off_q"], *scales)]
        xp, xx, yy, A, C, og, Q, oq, sp, sg, sq = leaves
        normal = torch.distributions.Normal
        value = (normal(xp @ A.t(), sp).log_prob(xx).sum(-1) +
                 normal(xx @ C.t() + og, sg).log_prob(yy.unsqueeze(1)).sum(-1) -
                 normal(xp @ Q.t() + oq.unsqueeze(1), sq).log_prob(xx).sum(-1))
        ref = torch.autograd.grad(value, leaves, grad_outputs=grad_lw.double())
        picks = (0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11)
        tolerance = 5e-5 if dtype == np.float32 else 1e-10
        for slot, want_grad in zip(picks, ref):
            scale = max(1.0, float(want_grad.abs().max()))
            assert float((got[slot].double() - want_grad).abs().max()) <= tolerance * scale, (tag, slot)
        grad = torch.from_numpy(rng.randn(B, K, dx).astype(dtype)).to(hip_device)
        gx, gw, goff = kernels.particle_affine_backward(grad, o["x_prev"], o["Q"], True, True, True)
        assert float((gx.double() - grad.double() @ o["Q"].double()).abs().max()) <= tolerance * 10, tag
        want_w = grad.double().reshape(-1, dx).t() @ o["x_prev"].double().reshape(-1, dx)
        assert float((gw.double() - want_w).abs().max()) <= tolerance * max(1.0, float(want_w.abs().max())), tag
        want_off = grad.double().sum(dim=1)
        assert float((goff.double() - want_off).abs().max()) <= tolerance * max(1.0, float(want_off.abs().max())), tag


@pytest.mark.parametrize("name", ["lgssm3d_smc_f64", "lgssm10d_smc_f64", "lgssm3d_smc_f32"])
def test_affine_route_reproduces_the_reference_s_own_fixtures(hip_device, name):
    """The fixtures captured from the imported reference (oracle/capture_golden.py: its `infer` on the
    d-dimensional LGSSM, every draw recorded) replayed through the SAME model stated with AffineNormal
    callables: float64 — every ancestor index of the reference, per-step log-weights and latents to
    1e-11, log Z to 1e-10, loss and parameter gradients; float32 — to float32 rounding where the indices
    agree (a CDF comparison within rounding noise may flip, as for any float32 implementation)."""
    from tests.golden_io import Golden
    from aesmc_amd import state as amd_state
    case = Golden(name)
    f64 = case.dtype == torch.float64
    parts, named = case.build_parts(amd_state, hip_device, affine=True)
    observations = case.observations(hip_device)
    launches = _count_affine_launches()
    with replay.replay(case.tape()), launches:
        result = inference.infer("smc", observations, parts["initial"], parts["transition"], parts["emission"],
                                 parts["proposal"], case.meta["num_particles"], return_log_marginal_likelihood=True,
                                 return_latents=True, return_original_latents=True, return_log_weights=True,
                                 return_ancestral_indices=True)
    steps = len(observations)
    assert launches.count["affine_rsample"] == steps - 1 and launches.count["affine_logweight"] == steps - 1
    got_idx = [a.cpu().numpy() for a in result["ancestral_indices"]]
    want_idx = case.series("out_idx")
    exact = all((g == w).all() for g, w in zip(got_idx, want_idx))
    if f64:
        assert exact
    else:
        assert np.mean([(g == w).mean() for g, w in zip(got_idx, want_idx)]) >= 0.999
    tol = dict(rtol=1e-11, atol=1e-11) if f64 else dict(rtol=1e-5, atol=1e-5)
    if exact:
        for got, want in zip(result["log_weights"], case.series("out_log_weights")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol)
        for got, want in zip(result["original_latents"], case.series("out_original_latents")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol)
        for got, want in zip(result["latents"], case.series("out_latents")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, **tol)
    lml, want = result["log_marginal_likelihood"].detach().cpu().numpy(), case["out_lml"]
    bound = ((1e-10 if f64 else 1e-4) if exact else 1e-2) * (1 + np.abs(want))
    assert (np.abs(lml - want) <= bound).all()
    with replay.replay(case.tape()):
        loss = losses.get_loss(observations, case.meta["num_particles"], "aesmc", parts["initial"],
                               parts["transition"], parts["emission"], parts["proposal"])
    loss.backward()
    assert abs(loss.item() - float(case["out_loss"])) <= ((1e-10 if f64 else 1e-4) if exact else 1e-2) * \
        (1 + abs(float(case["out_loss"])))
    if exact:
        for pname, parameter in named.items():
            want = case["grad_" + pname]
            scale = np.abs(want).max() + 1e-30
            np.testing.assert_allclose(parameter.grad.cpu().numpy() / scale, want / scale, rtol=0,
                                       atol=1e-8 if f64 else 1e-3)


def test_full_size_step_is_the_same_by_every_route(hip_device):
    """BASELINE.json's north-star shape (B=1024, K=4096, d=10; four timesteps): the run whose proposal
    defers its draw (K15 forward, K14 backward) against the run that draws at once (K9, K10, K14) and the
    one with matmul callables (PyTorch matmuls, K6 / K5, autograd): the first two identical in every bit —
    ancestors, latents, evidence, gradients — the third within float32 rounding: the first step's ancestor
    indices equal except where a CDF comparison sits inside that rounding, evidence and gradients close."""
    from aesmc_amd import inference
    from aesmc_amd.testing.models import LgssmNd
    B, K, T = 1024, 4096, 4
    runs = {}
    for name, kwargs in (("deferred", dict(affine=True, defer_draw=True)), ("immediate", dict(affine=True, defer_draw=False)),
                         ("matmul", dict(affine=False)), ("linked", dict(affine=True, defer_draw=True))):
        model = LgssmNd(10, dtype=torch.float32, validate_args=False, **kwargs).tune_proposal().to(hip_device)
        observations = model.simulate(T, B, seed=5)
        torch.manual_seed(21)
        np.random.seed(21)
        # "linked": consecutive steps' nodes hand the gather's backward on (the default; K14 adds a particle's
        # children in k order, the stand-alone segmented sum associates differently) — the bit-for-bit comparison
        # of the routes is made with every step launching that sum itself
        with inference.lazy_gather(name != "matmul"), inference.fold_gather_backward(name == "linked"):
            out = inference.infer("smc", observations, model.initial, model.transition, model.emission, model.proposal,
                                  K, return_log_marginal_likelihood=True, return_log_weight=False, return_latents=False,
                                  return_ancestral_indices=True)
        (-out["log_marginal_likelihood"].mean()).backward()
        runs[name] = (out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        del model, observations
    (a, ga), (b, gb), (c, gc) = runs["deferred"], runs["immediate"], runs["matmul"]
    assert torch.equal(a["log_marginal_likelihood"], b["log_marginal_likelihood"])
    assert torch.equal(a["last_latent"], b["last_latent"])
    for x, y in zip(a["ancestral_indices"], b["ancestral_indices"]):
        assert torch.equal(x, y)
    assert sorted(ga) == sorted(gb) and ga
    for name in ga:
        assert torch.equal(ga[name], gb[name]), name
    # the default route (linked step nodes): the same forward pass in every bit, gradients to float32 rounding
    d, gd = runs["linked"]
    assert torch.equal(a["log_marginal_likelihood"], d["log_marginal_likelihood"])
    assert torch.equal(a["last_latent"], d["last_latent"])
    assert sorted(ga) == sorted(gd)
    for name in ga:
        scale = max(float(ga[name].abs().max()), 1e-30)
        assert float((ga[name] - gd[name]).abs().max()) <= 2e-5 * scale, (name, "linked step nodes")
    # against the matmul statement of the same model: float32 rounding apart
    # (a flipped ancestor puts a different particle into that slot for good, so later steps differ in more places:
    # the first resampling step is the one bounded by CDF rounding alone)
    first = int((a["ancestral_indices"][0] != c["ancestral_indices"][0]).sum())
    assert first <= 2e-3 * B * K, first
    differing = sum(int((x != y).sum()) for x, y in zip(a["ancestral_indices"], c["ancestral_indices"]))
    assert differing <= 5e-2 * (T - 1) * B * K, differing
    torch.testing.assert_close(a["log_marginal_likelihood"], c["log_marginal_likelihood"], rtol=1e-3, atol=1e-2)
    assert sorted(ga) == sorted(gc)
    for name in ga:
        scale = max(float(gc[name].abs().max()), 1e-30)
        assert float((ga[name] - gc[name]).abs().max()) <= 5e-2 * scale, name


# ---- K13 / K13b: the proposal net and its backward --------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(3, 700, 10, 64, 10), (2, 513, 5, 16, 3), (5, 64, 16, 33, 16), (1, 1000, 3, 7, 7),
                                   (4, 300, 1, 1, 1), (128, 4096, 10, 64, 10)])
def test_particle_mlp_matches_the_c_oracle_and_the_pytorch_expression(kernels, hip_device, dtype, shape):
    """K13 against oracle/smc_core.c (same fma chains; the device's tanh against libm's: a few ulp) and
    against the PyTorch expression it replaces (cat + Linear + tanh + Linear)."""
    B, K, din, hid, dout = shape
    rng = np.random.RandomState(B + K + hid)
    r = lambda *s_: rng.randn(*s_).astype(dtype)
    x, w1, off1, w2, b2 = r(B, K, din), (r(hid, din) / np.sqrt(din)).astype(dtype), r(B, hid), \
        (r(dout, hid) / np.sqrt(hid)).astype(dtype), r(dout)
    dev = lambda a: torch.from_numpy(a).to(hip_device)
    out = kernels.particle_mlp(dev(x), dev(w1), dev(off1), dev(w2), dev(b2))
    assert out is not None and out.shape == (B, K, dout)
    tolerance = 3e-6 if dtype == np.float32 else 1e-14
    if B * K <= 4096:
        want = c_oracle.particle_mlp(x, w1, off1, w2, b2)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=tolerance, atol=tolerance * 4)
    ref = torch.tanh(dev(x).double() @ dev(w1).double().t() + dev(off1).double().unsqueeze(1)) @ dev(w2).double().t() \
        + dev(b2).double()
    assert float((out.double() - ref).abs().max()) <= 4 * tolerance * max(1.0, float(ref.abs().max()))
    shared = kernels.particle_mlp(dev(x), dev(w1), dev(off1[0]), dev(w2), None)     # [H] offset, no output bias
    ref = torch.tanh(dev(x).double() @ dev(w1).double().t() + dev(off1[0]).double()) @ dev(w2).double().t()
    assert float((shared.double() - ref).abs().max()) <= 4 * tolerance * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(3, 768, 10, 64, 10), (2, 512, 5, 16, 3), (5, 256, 15, 33, 16), (1, 1024, 3, 7, 7),
                                   (4, 256, 1, 1, 1), (128, 4096, 10, 64, 10)])
def test_particle_mlp_backward_equals_float64_autograd(kernels, hip_device, dtype, shape):
    """K13b — grad_x per particle, grad_W1 / grad_W2 contracted over the particles on the matrix cores (one record per
    wavefront, added by the binder), the per-row sums of dh riding in the column of ones — against float64 autograd of
    tanh(x W1^T + c1) W2^T: float64 to 1e-11 of a gradient's largest entry, float32 to rounding of sums over B K
    particles.  The last shape is configs[3]'s per-GPU shard (B = 128, K = 4096, d = 10, 64 hidden units)."""
    B, K, din, hid, dout = shape
    gen = torch.Generator(device=hip_device).manual_seed(B + K + hid)
    make = lambda *s_: torch.randn(*s_, device=hip_device, dtype=dtype, generator=gen)
    x, w1, off1, w2 = make(B, K, din), make(hid, din) / din ** 0.5, make(B, hid), make(dout, hid) / hid ** 0.5
    upstream = make(B, K, dout)
    got = kernels.particle_mlp_backward(upstream, x, w1, off1, w2)
    assert got is not None, "K13b declined a shape it is built for"
    leaves = [t.double().requires_grad_(True) for t in (x, w1, off1, w2)]
    out = torch.tanh(leaves[0] @ leaves[1].t() + leaves[2].unsqueeze(1)) @ leaves[3].t()
    want = torch.autograd.grad(out, leaves, upstream.double())
    # (float32: a sum over B K products of order one that cancels to a tenth of that — the 1 x 1 net — keeps ~5e-5)
    tolerance = 1e-11 if dtype == torch.float64 else 6e-5
    for name, a, b in zip(("grad_x", "grad_w1", "grad_offset1", "grad_w2"), got, want):
        assert a.shape == b.shape, (name, a.shape, b.shape)
        scale = max(float(b.abs().max()), 1e-30)
        assert float((a.double() - b).abs().max()) <= tolerance * scale, (name, float((a.double() - b).abs().max()), scale)
    # what it leaves to the caller's own operations: K not a multiple of 256, sixteen inputs (no column left for the ones)
    assert kernels.particle_mlp_backward(make(2, 300, dout), make(2, 300, din), w1, make(2, hid), w2) is None
    if din < 16:
        wide_in = make(2, 256, 16)
        assert kernels.particle_mlp_backward(make(2, 256, dout), wide_in, make(hid, 16), make(2, hid), w2) is None


def test_particle_mlp_operator_gradients_and_fallbacks(hip_device):
    from aesmc_amd.linear_gaussian import particle_mlp
    gen = torch.Generator(device=hip_device).manual_seed(4)
    make = lambda *shape: torch.randn(*shape, device=hip_device, dtype=torch.float64, generator=gen).requires_grad_(True)
    for K in (512, 300):      # a multiple of 256: K13b; else K13 forward and PyTorch's autograd of the expression
        x, w1, off1, w2, b2 = make(3, K, 6), make(20, 6), make(3, 20), make(4, 20), make(4)
        out = particle_mlp(x, w1, off1, w2, b2)
        upstream = torch.randn(out.shape, device=hip_device, dtype=torch.float64, generator=gen)
        got = torch.autograd.grad(out, (x, w1, off1, w2, b2), upstream)
        ref_out = torch.tanh(x @ w1.t() + off1.unsqueeze(1)) @ w2.t() + b2
        ref = torch.autograd.grad(ref_out, (x, w1, off1, w2, b2), upstream)
        torch.testing.assert_close(out, ref_out, rtol=1e-12, atol=1e-12)
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-10)
    shared = make(20)      # one offset for every row: its gradient is the sum over the rows
    out = particle_mlp(x.detach(), w1.detach(), shared, w2.detach(), None)
    (grad_shared,) = torch.autograd.grad(out, (shared,), torch.ones_like(out))
    ref = torch.autograd.grad(torch.tanh(x.detach() @ w1.detach().t() + shared) @ w2.detach().t(), (shared,),
                              torch.ones_like(out))[0]
    torch.testing.assert_close(grad_shared, ref, rtol=1e-10, atol=1e-10)
    wide = particle_mlp(make(2, 64, 6), make(100, 6), make(100), make(4, 100))     # beyond 64 hidden units: PyTorch
    assert wide.shape == (2, 64, 4)
    few = particle_mlp(make(50, 8, 6), w1, make(50, 20), w2, b2)                     # 8 particles per row: PyTorch
    assert few.shape == (50, 8, 4)


@pytest.mark.parametrize("algorithm,B,K,T,d", [("aesmc", 3, 256, 4, 5), ("aesmc", 2, 512, 3, 10)])
def test_fused_nonlinear_model_matches_the_cpu_port_with_gradients(hip_device, algorithm, B, K, T, d):
    """BASELINE.json's nonlinear state-space model with its maps through K8 and its proposal net through
    K13 / K13b (`fused=True`) against the CPU port running the plain PyTorch callables, draws replayed:
    float64 loss to 1e-10, every parameter gradient to 1e-8 of its largest entry."""
    from aesmc_amd import _kernels, losses
    from aesmc_amd.testing import models, replay
    from oracle import reference_port
    dtype = torch.float64
    cpu_model = models.NonlinearSsm(d, hidden=24, seed=0, dtype=dtype, state=reference_port)
    observations = cpu_model.simulate(T, B, seed=1)
    np.random.seed(5)
    torch.manual_seed(5)
    with replay.record() as tape:
        want = reference_port.get_loss(observations, K, algorithm, *_parts(cpu_model))
    want.backward()
    model = models.NonlinearSsm(d, hidden=24, seed=0, dtype=dtype, fused=True).to(hip_device)
    provider = _kernels.get()
    calls = {"forward": 0, "backward": 0}
    forward, backward = provider.particle_mlp, provider.particle_mlp_backward

    def spy_forward(*args, **kwargs):
        calls["forward"] += 1
        return forward(*args, **kwargs)

    def spy_backward(*args, **kwargs):
        out = backward(*args, **kwargs)
        calls["backward"] += out is not None
        return out
    provider.particle_mlp, provider.particle_mlp_backward = spy_forward, spy_backward
    try:
        with replay.replay(tape):
            got = losses.get_loss([o.to(hip_device) for o in observations], K, algorithm, *_parts(model))
        got.backward()
    finally:
        del provider.particle_mlp, provider.particle_mlp_backward
    assert calls == {"forward": T - 1, "backward": T - 1}, calls
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-10, atol=1e-10)
    expected = dict(cpu_model.named_parameters())
    for name, parameter in model.named_parameters():
        reference = expected[name].grad
        if reference is None:
            continue
        scale = max(float(reference.abs().max()), 1e-30)
        assert float((parameter.grad.cpu() - reference).abs().max()) <= 1e-8 * scale, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_particle_affine_through_tanh_is_torch_tanh_of_the_plain_launch(kernels, hip_device, dtype):
    """aesmc_particle_affine_tanh: the values are `torch.tanh` of what the plain launch stores — the
    same bits (the device library's tanh on the same chain) — and the operator's gradients are autograd's of
    `tanh(x @ W.t() + c)` in float64."""
    from aesmc_amd.linear_gaussian import particle_affine
    gen = torch.Generator(device=hip_device).manual_seed(3)
    make = lambda *s_: torch.randn(*s_, device=hip_device, dtype=dtype, generator=gen)
    x, W, c = make(5, 700, 10), make(10, 10) * 0.4, make(5, 10)
    plain = kernels.particle_affine(x, W, c)
    fused = kernels.particle_affine(x, W, c, through_tanh=True)
    assert torch.equal(fused, torch.tanh(plain))
    leaves = [t.clone().requires_grad_(True) for t in (x, W, c)]
    out = particle_affine(*leaves, activation="tanh")
    upstream = make(*out.shape)
    got = torch.autograd.grad(out, leaves, upstream)
    ref_leaves = [t.double().requires_grad_(True) for t in (x, W, c)]
    ref = torch.tanh(ref_leaves[0] @ ref_leaves[1].t() + ref_leaves[2].unsqueeze(1))
    want = torch.autograd.grad(ref, ref_leaves, upstream.double())
    tolerance = 1e-11 if dtype == torch.float64 else 3e-5
    for a, b in zip(got, want):
        assert float((a.double() - b).abs().max()) <= tolerance * max(1.0, float(b.abs().max()))
    wide = particle_affine(make(2, 64, 20), make(20, 20), activation="tanh")      # wider than K8 takes: PyTorch's operators
    assert wide.shape == (2, 64, 20)
    with pytest.raises(ValueError):
        particle_affine(x, W, c, activation="relu")
