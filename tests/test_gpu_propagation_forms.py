"""The forms of the fused propagation launch (aesmc_affine_normal_propagate_drawn: resampling gather, the proposal's
noise and draw, the step's log-weight — aesmc/inference.py:102-126, state.py:98, :179 for a linear-Gaussian model)
against each other and against `oracle/`:

  * the item form (linear_gaussian_item.hip: one work item per workgroup, what one GPU's shard of a batch takes) equals
    the persistent form (linear_gaussian_fused.hip; above 12 values per row: the first form, linear_gaussian_noise.hip) BIT
    FOR BIT — x_t and the log-weights — at the strong-scaling shard shapes of the north-star batch, at every latent extent
    2 .. 16 with equal and unequal observation extents, ragged K, windows that straddle batch rows, with and without the
    gather, healthy and collapsed ancestries;
  * x_t of the item form equals oracle/smc_core.c bit for bit, its log-weights to the tolerance the stand-alone
    log-weight kernel is held to;
  * bad ancestor indices are flagged, never followed.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from tests.test_gpu_linear_gaussian import operands
from tests.test_gpu_noise_and_lazy_latents import _ancestors

pytestmark = pytest.mark.gpu

PERSISTENT, ITEM = 1, 2


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


@pytest.fixture()
def forms(kernels, monkeypatch):
    """Pins the launch's form for the duration of a test: forms(PERSISTENT) / forms(ITEM); back to the policy afterwards."""
    lib = kernels._lib
    monkeypatch.setattr(type(kernels), "DRAWN_MIN_PARTICLES", 0)
    yield lambda form: lib.aesmc_test_set_k16_form(form)
    lib.aesmc_test_set_k16_form(0)


# (B, K, dx, dy): one GPU's shards of the north-star batch (B = 128, 256), configs[1]'s shape, every even extent with
# equal (compile-time) and unequal (run-time) observation extents, K ragged / prime / barely a window, one batch row
FORM_SHAPES = [(128, 4096, 10, 10), (256, 1024, 10, 10), (300, 4099, 10, 7), (37, 29000, 6, 9), (1, 128, 2, 2),
               (2, 4096, 12, 12), (64, 1024, 8, 4), (9, 513, 4, 1), (3, 200000, 10, 10), (7, 2222, 12, 5), (5, 131, 2, 12),
               (33, 640, 6, 6), (12, 4096, 8, 8), (4, 8192, 4, 4), (2, 130, 10, 10),
               # odd extents (rows of dwords) and rows of 13 .. 16 values
               (5, 777, 3, 11), (3, 300, 5, 5), (2, 1000, 7, 3), (4, 640, 9, 9), (3, 512, 11, 11), (2, 700, 13, 2),
               (2, 2048, 14, 14), (3, 1000, 15, 16), (6, 1024, 16, 16), (2, 4096, 16, 5), (130, 4096, 16, 16)]


def _run(kernels, o, x_prev, y, off_p, off_q, idx, seed):
    from aesmc_amd import _philox
    dev = x_prev.device
    B, K, dx = x_prev.shape
    torch.manual_seed(seed)
    reservation = _philox.reserve(B * K * dx, dev)
    out_x = torch.full_like(x_prev, float("nan"))
    lw = kernels.affine_propagate_drawn(x_prev, reservation, y, (o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], off_q),
                                        (o["s_p"], o["s_g"], o["s_q"]), out_x=out_x, ancestors=idx)
    assert lw is not None
    return out_x, lw, kernels._lib.aesmc_test_last_k16_form()


@pytest.mark.parametrize("gather", [True, False])
@pytest.mark.parametrize("shape", FORM_SHAPES)
def test_the_item_form_equals_the_persistent_form_bit_for_bit(kernels, hip_device, forms, shape, gather):
    B, K, dx, dy = shape
    _, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=B + K)
    gen = torch.Generator(device=hip_device).manual_seed(K + dx)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    off_p = torch.randn(dx, device=hip_device, generator=gen)
    for spread in ((1.0, 5.0) if gather else (None,)):
        idx = _ancestors(B, K, hip_device, seed=B + K, spread=spread) if gather else None
        kernels.read_flags(hip_device)
        forms(PERSISTENT)
        want_x, want_lw, ran = _run(kernels, o, x_prev, y, off_p, off_q, idx, seed=5 + K)
        assert ran == PERSISTENT
        forms(ITEM)
        got_x, got_lw, ran = _run(kernels, o, x_prev, y, off_p, off_q, idx, seed=5 + K)
        assert ran == ITEM, "the item form declined a shape it is built for"
        assert kernels.read_flags(hip_device) == 0
        assert torch.equal(got_x, want_x), float((got_x - want_x).abs().max())
        assert got_lw.cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes()


@pytest.mark.parametrize("shape", [(128, 4096, 10, 10), (37, 29000, 6, 9), (1, 128, 2, 2), (7, 2222, 12, 5), (9, 513, 4, 1),
                                   (64, 1024, 8, 4), (5, 777, 3, 11), (3, 512, 11, 11), (3, 1000, 15, 16), (6, 1024, 16, 16)])
def test_the_item_form_equals_the_c_oracle(kernels, hip_device, forms, shape):
    """x_t bit for bit (gather of the ancestor rows, one fma chain per element started from the offset, eps * s rounded
    before the sum, eps = what `torch.empty(shape).normal_()` holds for the same generator state), the log-weight to
    5e-7 relative (the device's log(sigma) against glibc's, times d)."""
    from aesmc_amd import _philox
    B, K, dx, dy = shape
    n, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=B + K)
    gen = torch.Generator(device=hip_device).manual_seed(K + dx)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    off_p = torch.randn(dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=1.0)
    torch.manual_seed(77 + K)
    torch.randn(5, device=hip_device)
    state = torch.cuda.get_rng_state(hip_device)
    eps = torch.empty(B, K, dx, device=hip_device).normal_()      # what the reference's rsample would draw (state.py:98)
    torch.cuda.set_rng_state(state, hip_device)
    reservation = _philox.reserve(B * K * dx, hip_device)
    forms(ITEM)
    got_x = torch.full_like(x_prev, float("nan"))
    got_lw = kernels.affine_propagate_drawn(x_prev, reservation, y, (o["A"], off_p), (o["C"], o["off_g"]), (o["Q"], off_q),
                                            (o["s_p"], o["s_g"], o["s_q"]), out_x=got_x, ancestors=idx)
    assert got_lw is not None and kernels._lib.aesmc_test_last_k16_form() == ITEM
    assert kernels.read_flags(hip_device) == 0
    moved, flags = c_oracle.gather(x_prev.cpu().numpy(), idx.cpu().numpy())
    assert flags == 0
    s_p, s_g, s_q = (float(o[k].cpu()) for k in ("s_p", "s_g", "s_q"))
    want_x = c_oracle.affine_rsample(moved, n["Q"], off_q.cpu().numpy(), eps.cpu().numpy(), s_q)
    np.testing.assert_array_equal(got_x.cpu().numpy(), want_x)
    want_lw = c_oracle.affine_logweight(moved, want_x, y.cpu().numpy(), (n["A"], off_p.cpu().numpy()),
                                        (n["C"], n["off_g"]), (n["Q"], off_q.cpu().numpy()), s_p, s_g, s_q)
    np.testing.assert_allclose(got_lw.cpu().numpy(), want_lw, rtol=5e-7,
                               atol=5e-7 * max(1.0, float(np.abs(want_lw).max())))


def test_the_item_form_flags_bad_ancestors_and_declines_what_it_does_not_cover(kernels, hip_device, forms, monkeypatch):
    B, K, dx, dy = 3, 1000, 10, 10
    _, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=1)
    gen = torch.Generator(device=hip_device).manual_seed(3)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=2, spread=1.0).clone()
    idx[0, 5], idx[2, 999], idx[1, 0] = K, -1, K + 7      # K is what K2 writes for a degenerate row; the others are nobody's
    forms(ITEM)
    kernels.read_flags(hip_device)
    out_x, lw, ran = _run(kernels, o, x_prev, y, None, off_q, idx, seed=9)
    assert ran == ITEM
    assert torch.isfinite(out_x).all() and torch.isfinite(lw).all()
    from aesmc_amd import _lib
    assert kernels.read_flags(hip_device) & _lib.FLAG_INDEX_OUT_OF_RANGE
    # strided weights (a transposed view): the item form takes them through the interleaved weight pairs (built from any
    # strides); without the pairs they are another form's — same call, same bits, other kernel
    for dx2, dy2 in ((10, 10), (6, 3)):
        _, o2 = operands(4, 32, dx2, dy2, np.float32, hip_device, seed=4)
        o2 = dict(o2, A=o2["A"].t().contiguous().t())
        x2 = torch.randn(2, 300, dx2, device=hip_device, generator=gen)
        y2 = torch.randn(2, dy2, device=hip_device, generator=gen)
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", False)
        want_x, want_lw, ran = _run(kernels, o2, x2, y2, None, None, None, seed=11)
        assert ran != ITEM and torch.isfinite(want_lw).all()
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", True)
        got_x, got_lw, ran = _run(kernels, o2, x2, y2, None, None, None, seed=11)
        assert ran == ITEM
        assert torch.equal(got_x, want_x) and got_lw.cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes()


@pytest.mark.parametrize("shape", [(128, 4096, 10, 10), (37, 29000, 6, 9), (5, 777, 3, 11), (9, 513, 4, 1), (3, 1000, 15, 16),
                                   (6, 1024, 16, 16), (2, 700, 13, 2), (1, 128, 2, 2), (64, 1024, 8, 4)])
def test_packed_multiply_adds_give_the_scalar_chains_bits(kernels, hip_device, forms, shape, monkeypatch):
    """The item form with its weights as interleaved pairs (one v_pk_fma_f32 advances the chains of two outputs) against
    the same form with one v_fmac_f32 per output and input: x_t and the log-weights bit for bit — odd and even extents,
    run-time and compile-time observation extents, an odd number of outputs (a pair's second row absent) — and the pairs
    follow the weights: changed in place, the next evaluation's launch sees the new values."""
    B, K, dx, dy = shape
    _, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=B + K)
    gen = torch.Generator(device=hip_device).manual_seed(K + dx)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=1.0)
    forms(ITEM)
    out = {}
    for paired in (False, True):
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", paired)
        x, lw, ran = _run(kernels, o, x_prev, y, None, off_q, idx, seed=5 + K)
        assert ran == ITEM
        out[paired] = (x, lw)
    assert torch.equal(out[True][0], out[False][0])
    assert out[True][1].cpu().numpy().tobytes() == out[False][1].cpu().numpy().tobytes()
    # the weights change in place (an optimiser step): same tensors, new values — the pairs are rebuilt
    with torch.no_grad():
        o["Q"].mul_(0.5)
        o["C"].add_(0.25)
    fresh = {}
    for paired in (False, True):
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", paired)
        fresh[paired] = _run(kernels, o, x_prev, y, None, off_q, idx, seed=5 + K)[:2]
    assert torch.equal(fresh[True][0], fresh[False][0]) and not torch.equal(fresh[True][0], out[True][0])
    assert fresh[True][1].cpu().numpy().tobytes() == fresh[False][1].cpu().numpy().tobytes()


@pytest.mark.parametrize("shape", [(16, 4096, 10, 10), (5, 777, 3, 11), (6, 1024, 16, 16), (9, 513, 4, 1)])
def test_the_densities_constants_behind_the_pairs_follow_the_scales(kernels, hip_device, forms, shape, monkeypatch):
    """aesmc_affine_weight_pairs_scaled leaves 2 s^2 and d (log s + log(2 pi) / 2) of the three densities behind the pairs
    and the item form reads them instead of taking three logarithms per wavefront: the log-weights' bits are those of
    the launch that forms them itself (pairs without constants, and no pairs at all); a scale changed in place, or
    other scale tensors, never meet stale constants; a model that hands in fresh scale tensors every timestep stops
    rebuilding after two launches.  (aesmc/state.py:98's Normal log-density, the same expression evaluated once.)"""
    from aesmc_amd import _philox
    B, K, dx, dy = shape
    _, o = operands(4, 32, dx, dy, np.float32, hip_device, seed=B + K)
    gen = torch.Generator(device=hip_device).manual_seed(K + dx)
    x_prev = torch.randn(B, K, dx, device=hip_device, generator=gen)
    y = torch.randn(B, dy, device=hip_device, generator=gen)
    off_q = torch.randn(B, dx, device=hip_device, generator=gen)
    idx = _ancestors(B, K, hip_device, seed=B + K, spread=1.0)
    forms(ITEM)
    kernels.begin_evaluation()

    def run(scales):
        return _run(kernels, dict(o, s_p=scales[0], s_g=scales[1], s_q=scales[2]), x_prev, y, None, off_q, idx, seed=5 + K)

    def tag():
        return int(kernels._pairs[1][-2:-1].view(torch.int32).item())

    scales = [o["s_p"], o["s_g"], o["s_q"]]
    monkeypatch.setattr(kernels, "WEIGHT_PAIRS", False)
    want_x, want_lw, ran = run(scales)
    assert ran == ITEM
    monkeypatch.setattr(kernels, "WEIGHT_PAIRS", True)
    got_x, got_lw, ran = run(scales)
    assert ran == ITEM and kernels._pairs[6] is not None and tag() == (0x5c000000 | (dy << 8) | dx)
    assert torch.equal(got_x, want_x) and got_lw.cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes()
    held = kernels._pairs[1]
    assert run(scales)[1].cpu().numpy().tobytes() == want_lw.cpu().numpy().tobytes() and kernels._pairs[1] is held
    # the constants are the kernel's own expressions of the scales
    consts = held[-8:-2].cpu().numpy()
    s = np.array([float(t.cpu()) for t in scales], dtype=np.float32)
    np.testing.assert_array_equal(consts[0::2], np.float32(2.0) * (s * s))
    extents = np.array([dx, dy, dx], dtype=np.float32)
    np.testing.assert_allclose(consts[1::2], extents * (np.log(s.astype(np.float64)) + 0.5 * np.log(2 * np.pi)), rtol=2e-6)
    # a scale changed in place: same tensor, new value, new constants
    with torch.no_grad():
        scales[1].mul_(1.75)
    monkeypatch.setattr(kernels, "WEIGHT_PAIRS", False)
    want_lw2 = run(scales)[1]
    monkeypatch.setattr(kernels, "WEIGHT_PAIRS", True)
    got_lw2 = run(scales)[1]
    assert kernels._pairs[1] is not held and kernels._pairs[7] == 1
    assert got_lw2.cpu().numpy().tobytes() == want_lw2.cpu().numpy().tobytes()
    assert got_lw2.cpu().numpy().tobytes() != want_lw.cpu().numpy().tobytes()
    # other scale tensors every call (a model that computes them per timestep): two rebuilds, then pairs without constants
    for trip in range(4):
        fresh = [t.clone() * (1.0 + 0.125 * trip) for t in scales]
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", False)
        want = run(fresh)[1]
        monkeypatch.setattr(kernels, "WEIGHT_PAIRS", True)
        got = run(fresh)[1]
        assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes(), trip
    assert kernels._pairs[6] is None and tag() == 0
    last = kernels._pairs[1]
    assert run([t.clone() for t in scales])[1] is not None and kernels._pairs[1] is last
    # the C ABI's two builders side by side: a cleared tag makes the launch form the constants itself
    import ctypes
    from aesmc_amd._kernels import _ptr
    maps = [kernels._affine_map(*term, slot=slot) for slot, term in enumerate(((o["A"], None), (o["C"], o["off_g"]), (o["Q"], off_q)))]
    outs = []
    for scaled in (False, True):
        pairs = kernels._build_pairs(maps, scales if scaled else None, hip_device)
        torch.manual_seed(5 + K)
        reservation = _philox.reserve(B * K * dx, hip_device)
        out_x = torch.empty_like(x_prev)
        lw = torch.empty(B, K, device=hip_device)
        status = kernels._lib.aesmc_affine_normal_propagate_drawn_paired(
            _ptr(x_prev), _ptr(idx), _ptr(y), y.stride(0), ctypes.byref(maps[0][0]), ctypes.byref(maps[1][0]),
            ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]), _ptr(scales[2]), _ptr(out_x), _ptr(lw),
            _ptr(kernels.flags(hip_device)), B, K, reservation.seed, reservation.offset, reservation.threads,
            _ptr(reservation.state), _ptr(pairs), kernels._stream(x_prev))
        assert status == 0
        outs.append((out_x, lw))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1].cpu().numpy().tobytes() == outs[1][1].cpu().numpy().tobytes()
    assert outs[1][1].cpu().numpy().tobytes() == want_lw2.cpu().numpy().tobytes()
    assert kernels.read_flags(hip_device) == 0
