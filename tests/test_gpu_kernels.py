"""GPU parity tests, kernel level: every C-ABI entry point (through aesmc_amd._kernels, which is a
ctypes veneer over include/aesmc_hip.h) against oracle/kernel_oracle.py on seeded inputs, against
the reference-captured fixtures in tests/golden/, and — at BASELINE.json's full sizes — through
size-independent properties.

Bars: integer / index / byte work bit-exact; float32 log-sum-exp within rtol 2e-6, float64 1e-13.
"""
import numpy as np
import pytest
import torch

from oracle import kernel_oracle
from tests.golden_io import Golden, RESAMPLER_CASES, float32_flip_bound, mismatch_margin, FLOAT32_CDF_NOISE

pytestmark = pytest.mark.gpu

F32_RTOL, F32_ATOL = 2e-6, 2e-6
F64_RTOL, F64_ATOL = 1e-13, 1e-13


def tol(dtype):
    return (F32_RTOL, F32_ATOL) if dtype in (torch.float32, np.float32) else (F64_RTOL, F64_ATOL)


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    provider.read_flags(hip_device)
    return provider


def dev(array, device):
    return torch.from_numpy(np.ascontiguousarray(array)).to(device)


# ---- K1 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (2, 16), (3, 7), (5, 64), (4, 1000), (7, 1024), (3, 1025),
                                   (2, 4096), (5, 8192), (300, 33), (2, 16385)])
def test_logweight_lse_matches_oracle(kernels, hip_device, dtype, shape):
    rng = np.random.RandomState(hash(shape) % 1000)
    a, b, c = [(3 * rng.randn(*shape)).astype(dtype) for _ in range(3)]
    lw, lse = kernels.logweight_lse(dev(a, hip_device), dev(b, hip_device), dev(c, hip_device))
    want_lw, want_lse = kernel_oracle.logweight_lse(a, b, c)
    np.testing.assert_array_equal(lw.cpu().numpy(), want_lw)  # two IEEE adds: exact
    rtol, atol = tol(dtype)
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_logweight_lse_optional_terms_and_special_values(kernels, hip_device, dtype):
    rng = np.random.RandomState(3)
    a = rng.randn(6, 40).astype(dtype)
    a[0, :] = -np.inf                 # empty row -> -inf
    a[1, 5] = np.inf                  # -> +inf
    a[2, 7] = np.nan                  # -> nan
    a[3, 3:30] = -np.inf              # partial -inf is fine
    a[4, :] = 1e4 if dtype == np.float64 else 80.0
    lw, lse = kernels.logweight_lse(dev(a, hip_device))
    want_lw, want_lse = kernel_oracle.logweight_lse(a)
    np.testing.assert_array_equal(lw.cpu().numpy(), want_lw)
    got = lse.cpu().numpy()
    assert got[0] == -np.inf and got[1] == np.inf and np.isnan(got[2])
    rtol, atol = tol(dtype)
    np.testing.assert_allclose(got[3:], want_lse[3:], rtol=rtol, atol=atol)
    # b only / c only
    b = rng.randn(6, 40).astype(dtype)
    lw_b, _ = kernels.logweight_lse(dev(a, hip_device), dev(b, hip_device), None)
    np.testing.assert_array_equal(lw_b.cpu().numpy()[3:], (a + b)[3:])
    lw_c, _ = kernels.logweight_lse(dev(a, hip_device), None, dev(b, hip_device))
    np.testing.assert_array_equal(lw_c.cpu().numpy()[3:], (a - b)[3:])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(2, 16), (3, 7), (4, 1000), (3, 2048), (2, 4100)])
def test_logweight_lse_backward_matches_oracle(kernels, hip_device, dtype, shape):
    rng = np.random.RandomState(5)
    lw = (2 * rng.randn(*shape)).astype(dtype)
    _, lse = kernel_oracle.logweight_lse(lw)
    glw = rng.randn(*shape).astype(dtype)
    glse = rng.randn(shape[0]).astype(dtype)
    rtol, atol = tol(dtype)
    for use_glw, use_glse in [(True, True), (False, True), (True, False)]:
        g, ng = kernels.logweight_lse_backward(
            dev(lw, hip_device), dev(lse, hip_device), dev(glw, hip_device) if use_glw else None,
            dev(glse, hip_device) if use_glse else None)
        want_g, want_ng = kernel_oracle.logweight_lse_backward(
            lw, lse, glw if use_glw else None, glse if use_glse else None)
        np.testing.assert_allclose(g.cpu().numpy(), want_g, rtol=10 * rtol, atol=10 * atol)
        np.testing.assert_array_equal(ng.cpu().numpy(), -g.cpu().numpy())


# ---- K2 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", RESAMPLER_CASES)
def test_ancestor_index_golden(kernels, hip_device, name):
    """Fixtures captured from aesmc.inference.sample_ancestral_index (reference)."""
    case = Golden(name)
    log_w, u = case["log_weight"], case["uniform"].reshape(-1)
    kernels.read_flags(hip_device)
    idx = kernels.ancestor_index(dev(log_w, hip_device), dev(u, hip_device)).cpu().numpy()
    flags = kernels.read_flags(hip_device)
    want, want_flags = kernel_oracle.ancestor_index(log_w, u)
    np.testing.assert_array_equal(idx, want)            # HIP == contract oracle, always
    assert flags == want_flags
    mismatches = int((idx != case["out_idx"]).sum())     # HIP vs the reference itself
    assert mismatches == case.meta["mismatches_vs_float64_cdf"]
    if log_w.dtype == np.float64:
        assert mismatches == 0
    else:
        assert mismatch_margin(log_w, u, idx, case["out_idx"]) <= FLOAT32_CDF_NOISE
        assert mismatches <= float32_flip_bound(log_w.shape[1]) * idx.size


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,scale", [((1, 1), 1.0), ((2, 16), 1.0), ((3, 7), 1.0), ((5, 64), 3.0),
                                         ((4, 129), 1.0), ((6, 1000), 1.0), ((3, 1024), 5.0),
                                         ((2, 4096), 1.0), ((2, 8191), 2.0), ((2, 16384), 1.0),
                                         ((300, 50), 1.0)])
def test_ancestor_index_matches_oracle(kernels, hip_device, dtype, shape, scale):
    rng = np.random.RandomState(shape[1])
    log_w = (scale * rng.randn(*shape)).astype(dtype)
    u = rng.uniform(size=shape[0])
    idx = kernels.ancestor_index(dev(log_w, hip_device), dev(u, hip_device)).cpu().numpy()
    want, _ = kernel_oracle.ancestor_index(log_w, u)
    np.testing.assert_array_equal(idx, want)
    assert (np.diff(idx, axis=1) >= 0).all()            # systematic resampling is monotone
    assert idx.min() >= 0 and idx.max() < shape[1]


def test_ancestor_index_randomised_stress(kernels, hip_device):
    """Random shapes and weight profiles aimed at the inverted search: ties (quantised weights),
    zero-weight stretches (-inf), one dominant particle, near-uniform weights whose CDF steps
    land exactly on positions (u = 0 with K a power of two), K around every chunk-size switch."""
    rng = np.random.RandomState(1234)
    sizes = [1, 2, 3, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 5000, 8192, 8193, 16384, 16385, 20000]
    for case in range(60):
        K = int(sizes[case % len(sizes)]) if case < 2 * len(sizes) else int(rng.randint(1, 3000))
        B = int(rng.randint(1, 5))
        dtype = [np.float32, np.float64][case % 2]
        kind = case % 6
        log_w = rng.randn(B, K) * [0.0, 0.1, 1.0, 10.0, 100.0, 1.0][kind]
        if kind == 5:
            log_w = np.round(log_w * 2) / 2                       # many exact ties
        if case % 4 == 1 and K > 4:
            lo = rng.randint(0, K - 2)
            log_w[:, lo:lo + max(1, K // 3)] = -np.inf             # a stretch without weight
            log_w[:, (lo + K // 2) % K] = 0.0                        # ... but never a dead row
        if case % 7 == 3:
            log_w[:, rng.randint(0, K)] += 50.0                      # one dominant particle
        log_w = log_w.astype(dtype)
        u = rng.uniform(size=B)
        if kind == 0:
            u[0] = 0.0                                               # positions k / K hit CDF steps exactly
        got = kernels.ancestor_index(dev(log_w, hip_device), dev(u, hip_device)).cpu().numpy()
        want, _ = kernel_oracle.ancestor_index(log_w, u)
        bad = np.argwhere(got != want)
        assert bad.size == 0, (case, K, B, dtype, kind, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
        # the same rows through the fused step, with a payload whose row length varies with the case
        # (incl. lengths the launch declines: then it must say so, not write something else)
        from oracle import c_oracle
        np.testing.assert_array_equal(c_oracle.ancestor_index(log_w, u)[0], want)
        d = [1, 2, 3, 4, 5, 8, 10, 12][case % 8]
        payload_dtype = [np.float32, np.float64, np.int32][case % 3]
        payload = (rng.randn(B, K, d) * 100).astype(payload_dtype)
        step = kernels.resample_step(dev(log_w, hip_device), dev(u, hip_device), dev(payload, hip_device),
                                     want_lse=True)
        if step is None:
            assert K > kernels.lds_max_particles or (K * d * payload.dtype.itemsize) % 16 != 0
            continue
        idx, lse, moved = step
        np.testing.assert_array_equal(idx.cpu().numpy(), want)
        np.testing.assert_array_equal(moved.cpu().numpy(), c_oracle.gather(payload, want)[0])
        _, want_lse = c_oracle.logweight_lse(log_w)
        rtol, atol = tol(dtype)
        np.testing.assert_allclose(lse.cpu().numpy().astype(np.float64), want_lse, rtol=rtol, atol=atol)


def test_ancestor_index_large_k_uses_workspace(kernels, hip_device):
    """K above the LDS-resident limit goes through the global-workspace CDF."""
    limit = int(kernels._lib.aesmc_ancestor_index_lds_max_particles())
    K = limit + 1500
    rng = np.random.RandomState(9)
    log_w = rng.randn(3, K)
    u = rng.uniform(size=3)
    assert kernels._lib.aesmc_workspace_bytes(3, K) == 3 * (K + K // 8 + 1) * 8   # padded float64 CDF
    idx = kernels.ancestor_index(dev(log_w, hip_device), dev(u, hip_device)).cpu().numpy()
    want, _ = kernel_oracle.ancestor_index(log_w, u)
    np.testing.assert_array_equal(idx, want)


def test_ancestor_index_flags(kernels, hip_device):
    rng = np.random.RandomState(2)
    log_w = rng.randn(4, 50).astype(np.float32)
    u = rng.uniform(size=4)
    kernels.read_flags(hip_device)
    kernels.ancestor_index(dev(log_w, hip_device), dev(u, hip_device))
    assert kernels.read_flags(hip_device) == 0
    bad = log_w.copy()
    bad[2, 7] = np.nan
    idx = kernels.ancestor_index(dev(bad, hip_device), dev(u, hip_device)).cpu().numpy()
    assert kernels.read_flags(hip_device) == kernel_oracle.FLAG_NAN_LOG_WEIGHT
    assert (idx[2] == 50).all()
    empty = log_w.copy()
    empty[1, :] = -np.inf
    idx = kernels.ancestor_index(dev(empty, hip_device), dev(u, hip_device)).cpu().numpy()
    assert kernels.read_flags(hip_device) == kernel_oracle.FLAG_DEGENERATE_ROW
    assert (idx[1] == 50).all() and idx[0].max() < 50


def test_ancestor_index_frequencies(kernels, hip_device):
    """The reference's statistical test (test/test_inference.py:64-84): weights [0.2, 0.3, 0.5],
    10 000 rows, empirical ancestor frequencies within 1e-2."""
    weight = np.array([0.2, 0.3, 0.5])
    trials = 10000
    rng = np.random.RandomState(0)
    log_w = np.log(np.broadcast_to(weight, (trials, 3))).astype(np.float32)
    idx = kernels.ancestor_index(dev(log_w, hip_device), dev(rng.uniform(size=trials), hip_device))
    idx = idx.cpu().numpy()
    freq = np.array([(idx == i).sum() for i in range(3)]) / (trials * 3)
    np.testing.assert_allclose(freq, weight, atol=1e-2)


# ---- K3 ------------------------------------------------------------------------------------------
def sorted_indices(rng, B, K, scale=1.0):
    log_w = scale * rng.randn(B, K)
    idx, _ = kernel_oracle.ancestor_index(log_w, rng.uniform(size=B))
    return idx


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64, np.uint8, np.float16, np.int16])
@pytest.mark.parametrize("shape", [(2, 16), (2, 16, 1), (3, 7, 3), (4, 100, 10), (2, 256, 128),
                                   (3, 33, 2, 5), (1, 1, 4), (5, 64, 7), (2, 1000, 6)])
def test_gather_matches_oracle(kernels, hip_device, dtype, shape):
    rng = np.random.RandomState(len(shape) * 100 + shape[1])
    src = (100 * rng.randn(*shape)).astype(dtype)
    for idx in (sorted_indices(rng, shape[0], shape[1]),
                rng.randint(0, shape[1], size=shape[:2]).astype(np.int64)):   # arbitrary order too
        out = kernels.gather(dev(src, hip_device), dev(idx, hip_device)).cpu().numpy()
        want, _ = kernel_oracle.gather(src, idx)
        np.testing.assert_array_equal(out, want)


def test_gather_reference_known_answers(kernels, hip_device):
    """test/test_state.py:286-303 (exact small gather)."""
    idx = torch.tensor([[1, 2, 0], [0, 0, 1]], device=hip_device)
    value = torch.tensor([[1., 2., 3.], [4., 5., 6.]], device=hip_device)
    want = torch.tensor([[2., 3., 1.], [4., 4., 5.]], device=hip_device)
    assert torch.equal(kernels.gather(value, idx), want)


def test_gather_strided_sources(kernels, hip_device):
    """Transposed time-0 latent (aesmc/state.py:102-103), stride-0 expands, sliced parents."""
    rng = np.random.RandomState(4)
    B, K, d = 5, 48, 3
    idx = sorted_indices(rng, B, K)
    idx_d = dev(idx, hip_device)
    for src in (torch.randn(K, B, device=hip_device).transpose(0, 1),
                torch.randn(K, B, d, device=hip_device).transpose(0, 1),
                torch.randn(B, 1, d, device=hip_device).expand(B, K, d),
                torch.randn(B, 2 * K, d, device=hip_device)[:, ::2],
                torch.randn(B, K, 2 * d, device=hip_device)[:, :, ::2],
                torch.randn(B, K, d + 1, device=hip_device)[:, :, 1:]):
        out = kernels.gather(src, idx_d)
        want, _ = kernel_oracle.gather(src.cpu().numpy(), idx)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        assert out.is_contiguous()


def test_gather_out_of_range_is_flagged_not_fatal(kernels, hip_device):
    src = torch.arange(24, dtype=torch.float32, device=hip_device).reshape(2, 4, 3)
    idx = torch.tensor([[0, 1, 4, 3], [-1, 0, 0, 2]], device=hip_device)
    kernels.read_flags(hip_device)
    out = kernels.gather(src, idx).cpu().numpy()
    assert kernels.read_flags(hip_device) == kernel_oracle.FLAG_INDEX_OUT_OF_RANGE
    want, flags = kernel_oracle.gather(src.cpu().numpy(), idx.cpu().numpy())
    assert flags == kernel_oracle.FLAG_INDEX_OUT_OF_RANGE
    np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,scale", [((2, 16), 1.0), ((3, 7, 3), 1.0), ((4, 100, 10), 1.0),
                                         ((2, 300, 10), 8.0), ((2, 1000, 6), 30.0), ((3, 64, 2, 5), 1.0)])
def test_gather_backward_matches_oracle(kernels, hip_device, dtype, shape, scale):
    rng = np.random.RandomState(shape[1])
    go = rng.randn(*shape).astype(dtype)
    for idx in (sorted_indices(rng, shape[0], shape[1], scale),
                rng.randint(0, shape[1], size=shape[:2]).astype(np.int64)):
        got = kernels.gather_backward(dev(go, hip_device), dev(idx, hip_device)).cpu().numpy()
        want, _ = kernel_oracle.gather_backward(go, idx)
        rtol, atol = tol(dtype)
        np.testing.assert_allclose(got, want, rtol=50 * rtol, atol=50 * atol)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,scale", [((2, 16), 1.0), ((3, 7, 3), 1.0), ((4, 100, 10), 1.0),
                                         ((2, 300, 10), 8.0), ((2, 1000, 6), 30.0), ((3, 64, 2, 5), 1.0),
                                         ((1, 5000, 10), 1.0), ((1, 5000, 10), 40.0), ((3, 700, 128), 3.0),
                                         ((2, 2049, 1), 2.0), ((300, 33, 4), 1.0), ((2, 600, 129), 0.0),
                                         ((1, 1, 7), 1.0), ((5, 1024, 16), 200.0)])
def test_gather_backward_sorted_path_matches_oracle(kernels, hip_device, dtype, shape, scale):
    """The segmented-sum backward (no atomics) on what K2 produces: uniform weights (scale 0:
    every particle survives), ordinary, and collapsed (scale >> 1: a few long runs crossing tiles
    and sections)."""
    rng = np.random.RandomState(shape[1] + int(scale))
    go = rng.randn(*shape).astype(dtype)
    idx = sorted_indices(rng, shape[0], shape[1], scale)
    kernels.read_flags(hip_device)
    got = kernels.gather_backward(dev(go, hip_device), dev(idx, hip_device), sorted_index=True).cpu().numpy()
    assert kernels.read_flags(hip_device) == 0
    want, _ = kernel_oracle.gather_backward(go, idx)
    rtol, atol = tol(dtype)
    longest = np.max(np.bincount(idx.reshape(-1) + np.repeat(np.arange(shape[0]), shape[1]) * shape[1]))
    # sums of `longest` standard-normal terms in a different association than the oracle's
    np.testing.assert_allclose(got, want, rtol=50 * rtol, atol=20 * atol * max(longest, 1) ** 0.5)
    # deterministic: same bits on a second launch (the atomic path cannot promise this)
    again = kernels.gather_backward(dev(go, hip_device), dev(idx, hip_device), sorted_index=True).cpu().numpy()
    np.testing.assert_array_equal(got, again)


def test_gather_backward_sorted_path_edge_indices(kernels, hip_device):
    K, d = 700, 3
    rng = np.random.RandomState(0)
    go = rng.randn(4, K, d)
    idx = np.stack([np.zeros(K, np.int64),                          # everyone descends from particle 0
                    np.full(K, K - 1, np.int64),                    # ... from the last particle
                    np.arange(K, dtype=np.int64),                   # identity
                    np.sort(rng.randint(0, K, size=K)).astype(np.int64)])
    got = kernels.gather_backward(dev(go, hip_device), dev(idx, hip_device), sorted_index=True).cpu().numpy()
    want, _ = kernel_oracle.gather_backward(go, idx)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-11)
    # a descent in an index promised sorted is reported, not silently mis-summed
    bad = idx.copy()
    bad[3, 300] = 0
    kernels.read_flags(hip_device)
    kernels.gather_backward(dev(go, hip_device), dev(bad, hip_device), sorted_index=True)
    assert kernels.read_flags(hip_device) & 16
    # the autograd wrapper only trusts indices tagged by K2 itself
    from aesmc_amd import _ops
    value = torch.randn(2, 50, 3, device=hip_device, dtype=torch.float64, requires_grad=True)
    tagged = _ops.ancestor_index(torch.randn(2, 50, device=hip_device), torch.rand(2, device=hip_device, dtype=torch.float64))
    assert getattr(tagged, "_aesmc_sorted", False)
    untagged = torch.randint(0, 50, (2, 50), device=hip_device)
    for index in (tagged, untagged):
        value.grad = None
        weights = torch.randn(2, 50, 3, device=hip_device, dtype=torch.float64)
        (_ops.resample_gather(value, index) * weights).sum().backward()
        want = torch.zeros_like(value).scatter_add_(1, index[..., None].expand(2, 50, 3), weights)
        torch.testing.assert_close(value.grad, want, rtol=1e-12, atol=1e-12)
    assert kernels.read_flags(hip_device) == 0


# ---- full-size properties (BASELINE.json configs) -----------------------------------------------
@pytest.mark.parametrize("B,K,d", [(256, 1024, 10), (1024, 4096, 10), (16, 16384, 128)])
def test_full_size_resample_properties(kernels, hip_device, B, K, d):
    gen = torch.Generator(device=hip_device).manual_seed(B + K)
    log_w = torch.randn(B, K, device=hip_device, generator=gen)
    u = torch.rand(B, device=hip_device, generator=gen, dtype=torch.float64)
    kernels.read_flags(hip_device)
    idx = kernels.ancestor_index(log_w, u)
    assert kernels.read_flags(hip_device) == 0
    assert bool((idx[:, 1:] >= idx[:, :-1]).all()) and int(idx.min()) >= 0 and int(idx.max()) < K
    # offspring counts are within 1 of K * normalised weight (systematic resampling's guarantee)
    counts = torch.zeros(B, K, device=hip_device, dtype=torch.float64)
    counts.scatter_add_(1, idx, torch.ones(B, K, device=hip_device, dtype=torch.float64))
    expected = torch.softmax(log_w.double(), dim=1) * K
    assert float((counts - expected).abs().max()) < 1.0 + 1e-6
    # a host-side check of a few rows against the oracle
    rows = [0, B // 2, B - 1]
    want, _ = kernel_oracle.ancestor_index(log_w[rows].cpu().numpy(), u[rows].cpu().numpy())
    np.testing.assert_array_equal(idx[rows].cpu().numpy(), want)
    # gather: equals torch.gather; identity index is a copy; checksum of gathered rows
    value = torch.randn(B, K, d, device=hip_device, generator=gen)
    out = kernels.gather(value, idx)
    assert torch.equal(out, torch.gather(value, 1, idx.unsqueeze(-1).expand_as(value)))
    identity = torch.arange(K, device=hip_device).unsqueeze(0).expand(B, K).contiguous()
    assert torch.equal(kernels.gather(value, identity), value)
    # the fused step gives the same indices and rows in one launch, and K1's row log-sum-exp
    step = kernels.resample_step(log_w, u, value, want_lse=True)
    assert torch.equal(step[0], idx) and torch.equal(step[2], out)
    torch.testing.assert_close(step[1], torch.logsumexp(log_w, dim=1), rtol=F32_RTOL, atol=F32_ATOL)
    # backward of gather is the adjoint: <gather(v), g> == <v, gather_backward(g)>
    g = torch.randn(B, K, d, device=hip_device, generator=gen)
    lhs = (out.double() * g.double()).sum()
    rhs = (value.double() * kernels.gather_backward(g, idx).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-5 * float(lhs.abs() + 1)


def test_full_size_logweight_lse(kernels, hip_device):
    """Config 3 of BASELINE.json: B=4096, K=8192 row log-sum-exp."""
    B, K = 4096, 8192
    gen = torch.Generator(device=hip_device).manual_seed(1)
    a, b, c = [torch.randn(B, K, device=hip_device, generator=gen) for _ in range(3)]
    lw, lse = kernels.logweight_lse(a, b, c)
    assert torch.equal(lw, a + b - c)
    want = torch.logsumexp(lw.double(), dim=1)
    assert float((lse.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    # shift invariance: lse(x + s) == lse(x) + s
    _, shifted = kernels.logweight_lse(lw + 3.0)
    assert float((shifted - (lse + 3.0)).abs().max()) < 1e-5


# ---- K4 ------------------------------------------------------------------------------------------
def _normal_case(rng, shape, dtype, loc_kind, scale_kind, device):
    """value plus loc / scale given as tensors of the shapes the three BatchShapeModes produce."""
    B, K = shape[:2]
    trail = shape[2:]
    value = torch.from_numpy(rng.randn(*shape).astype(dtype)).to(device)
    loc_shape = {"full": shape, "batch": (B, 1) + trail, "none": trail, "scalar": ()}[loc_kind]
    scale_shape = {"full": shape, "batch": (B, 1) + trail, "none": trail, "scalar": ()}[scale_kind]
    loc = torch.from_numpy(np.asarray(rng.randn(*loc_shape)).astype(dtype)).to(device)
    scale = torch.from_numpy(np.asarray(0.3 + rng.rand(*scale_shape)).astype(dtype)).to(device)
    return value, loc, scale


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(2, 16), (3, 7, 1), (4, 100, 10), (2, 33, 3, 5), (2, 50, 64), (2, 9, 65),
                                   (3, 20, 128), (1, 1, 4), (2, 300, 16), (2, 64, 32)])
@pytest.mark.parametrize("loc_kind,scale_kind", [("full", "scalar"), ("full", "full"), ("batch", "none"),
                                                 ("none", "scalar"), ("scalar", "batch")])
def test_normal_logprob_sum_matches_oracle_and_torch(kernels, hip_device, dtype, shape, loc_kind, scale_kind):
    rng = np.random.RandomState(len(shape) * 7 + shape[1])
    value, loc, scale = _normal_case(rng, shape, dtype, loc_kind, scale_kind, hip_device)
    got = kernels.normal_logprob_sum(value, loc.expand(shape), scale.expand(shape)).cpu().numpy()
    want = kernel_oracle.normal_logprob_sum(value.cpu().numpy(), loc.cpu().numpy(), scale.cpu().numpy())
    eager = torch.distributions.Normal(loc, scale).log_prob(value).reshape(shape[0], shape[1], -1).sum(2)
    D = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    rtol = (4e-6 if dtype == np.float32 else 1e-13) * max(1, D) ** 0.5
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * D)
    np.testing.assert_allclose(got, eager.cpu().numpy(), rtol=rtol, atol=rtol * D)
    if D == 1:  # no reduction: element arithmetic is PyTorch's own order, bit for bit
        np.testing.assert_array_equal(got, eager.cpu().numpy())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(2, 16), (4, 100, 10), (2, 33, 3, 5), (2, 9, 65), (3, 20, 128)])
@pytest.mark.parametrize("loc_kind,scale_kind", [("full", "scalar"), ("full", "full"), ("batch", "none")])
def test_normal_logprob_sum_backward_matches_oracle(kernels, hip_device, dtype, shape, loc_kind, scale_kind):
    rng = np.random.RandomState(shape[1])
    value, loc, scale = _normal_case(rng, shape, dtype, loc_kind, scale_kind, hip_device)
    go = torch.from_numpy(rng.randn(*shape[:2]).astype(dtype)).to(hip_device)
    got = kernels.normal_logprob_sum_backward(value, loc.expand(shape), scale.expand(shape), go, True, True, True)
    want = kernel_oracle.normal_logprob_sum_backward(value.cpu().numpy(), loc.cpu().numpy(),
                                                     scale.cpu().numpy(), go.cpu().numpy())
    rtol = 1e-5 if dtype == np.float32 else 1e-12
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w, rtol=rtol, atol=rtol)
    only_loc = kernels.normal_logprob_sum_backward(value, loc.expand(shape), scale.expand(shape), go,
                                                   False, True, False)
    assert only_loc[0] is None and only_loc[2] is None
    np.testing.assert_array_equal(only_loc[1].cpu().numpy(), got[1].cpu().numpy())


def test_normal_logprob_sum_strided_views(kernels, hip_device):
    """Operands as the SMC loop really produces them: transposed time-0 latent, observation
    expanded over particles, per-dimension scale vector."""
    B, K, d = 4, 37, 10
    gen = torch.Generator(device=hip_device).manual_seed(0)
    latent = torch.randn(K, B, d, device=hip_device, generator=gen).transpose(0, 1)     # state.py:102-103
    loc_b = torch.randn(B, d, device=hip_device, generator=gen)
    obs = torch.randn(B, d, device=hip_device, generator=gen).unsqueeze(1).expand(B, K, d)  # state.py:202
    loc_full = torch.randn(B, K, d, device=hip_device, generator=gen)
    scale_vec = 0.5 + torch.rand(d, device=hip_device, generator=gen)
    for value, loc, scale in [(latent, loc_b.unsqueeze(1), torch.tensor(0.7, device=hip_device)),
                              (obs, loc_full, scale_vec),
                              (loc_full[:, 0:36:2], loc_full[:, 1:36:2], scale_vec)]:
        shape = value.shape
        got = kernels.normal_logprob_sum(value, loc.expand(shape), scale.expand(shape))
        want = torch.distributions.Normal(loc, scale).log_prob(value).sum(-1)
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,K,dx,dy", [(2, 16, 1, 1), (3, 37, 1, 1), (2, 1030, 1, 1), (3, 4096, 1, 1), (3, 37, 10, 10), (4, 300, 10, 4), (2, 1000, 3, 7),
                                       (1, 257, 64, 64), (5, 64, 33, 2), (2, 4096, 10, 10)])
def test_normal_logweight_is_bitwise_the_unfused_route(kernels, hip_device, dtype, B, K, dx, dy):
    """K5 == K4 x 3 combined by K1, bit for bit, for the operand layouts the SMC loop produces."""
    gen = torch.Generator(device=hip_device).manual_seed(B * K + dx)

    def rand(*shape):
        return torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)

    scales = [torch.tensor(v, device=hip_device, dtype=dtype) for v in (1.0, 0.5, 0.7)]
    layouts = [
        # t > 0: dense latent, FULLY_EXPANDED transition / emission / proposal, observation expanded
        dict(x=rand(B, K, dx), loc_p=rand(B, K, dx), y=rand(B, dy).unsqueeze(1).expand(B, K, dy),
             loc_g=rand(B, K, dy), loc_q=rand(B, K, dx)),
        # t = 0: transposed latent, NOT_EXPANDED prior, BATCH_EXPANDED proposal
        dict(x=rand(K, B, dx).transpose(0, 1), loc_p=rand(dx).expand(B, K, dx),
             y=rand(B, dy).unsqueeze(1).expand(B, K, dy), loc_g=rand(B, K, dy),
             loc_q=rand(B, dx).unsqueeze(1).expand(B, K, dx)),
        # sliced (non-dense) operands
        dict(x=rand(B, 2 * K, dx)[:, ::2], loc_p=rand(B, K, dx + 1)[:, :, 1:], y=rand(B, K, dy),
             loc_g=rand(B, K, dy), loc_q=rand(B, K, dx)),
    ]
    for lay in layouts:
        sp, sg, sq = [s.expand(lay[k].shape) for s, k in zip(scales, ("x", "y", "x"))]
        fused = kernels.normal_logweight(lay["x"], lay["loc_p"], sp, lay["y"], lay["loc_g"], sg, lay["loc_q"], sq)
        assert fused is not None
        log_p = kernels.normal_logprob_sum(lay["x"], lay["loc_p"], sp)
        log_g = kernels.normal_logprob_sum(lay["y"], lay["loc_g"], sg)
        log_q = kernels.normal_logprob_sum(lay["x"], lay["loc_q"], sq)
        unfused, _ = kernels.logweight_lse(log_p, log_g, log_q)
        assert torch.equal(fused, unfused)
        eager = (torch.distributions.Normal(lay["loc_p"], scales[0]).log_prob(lay["x"]).sum(-1)
                 + torch.distributions.Normal(lay["loc_g"], scales[1]).log_prob(lay["y"]).sum(-1)
                 - torch.distributions.Normal(lay["loc_q"], scales[2]).log_prob(lay["x"]).sum(-1))
        tol_ = 2e-5 if dtype == torch.float32 else 1e-12
        torch.testing.assert_close(fused, eager, rtol=tol_, atol=tol_ * (dx + dy))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,K,dx,dy", [(2, 50, 128, 128), (1, 33, 68, 72), (3, 7, 256, 512), (2, 16, 1024, 1024)])
def test_normal_logweight_wide_rows_are_bitwise_the_unfused_route(kernels, hip_device, dtype, B, K, dx, dy):
    """More than 64 values per particle (d = 128 of BASELINE.json configs[4]): the row kernel."""
    gen = torch.Generator(device=hip_device).manual_seed(B * K + dx)
    rand = lambda *shape: torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)
    scales = [torch.tensor(v, device=hip_device, dtype=dtype) for v in (1.0, 0.5, 0.7)]
    layouts = [
        dict(x=rand(B, K, dx), loc_p=rand(B, K, dx), y=rand(B, dy).unsqueeze(1).expand(B, K, dy),
             loc_g=rand(B, K, dy), loc_q=rand(B, K, dx)),
        dict(x=rand(K, B, dx).transpose(0, 1), loc_p=rand(dx).expand(B, K, dx),
             y=rand(B, dy).unsqueeze(1).expand(B, K, dy), loc_g=rand(B, K, dy),
             loc_q=rand(B, dx).unsqueeze(1).expand(B, K, dx)),
        dict(x=rand(B, 2 * K, dx)[:, ::2], loc_p=rand(B, K, 2 * dx)[:, :, dx:], y=rand(B, K, dy),
             loc_g=rand(B, K, dy), loc_q=rand(B, K, dx)),
    ]
    for lay in layouts:
        sp, sg, sq = [s.expand(lay[k].shape) for s, k in zip(scales, ("x", "y", "x"))]
        fused = kernels.normal_logweight(lay["x"], lay["loc_p"], sp, lay["y"], lay["loc_g"], sg, lay["loc_q"], sq)
        assert fused is not None
        log_p = kernels.normal_logprob_sum(lay["x"], lay["loc_p"], sp)
        log_g = kernels.normal_logprob_sum(lay["y"], lay["loc_g"], sg)
        log_q = kernels.normal_logprob_sum(lay["x"], lay["loc_q"], sq)
        unfused, _ = kernels.logweight_lse(log_p, log_g, log_q)
        assert torch.equal(fused, unfused)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,K,dx,dy", [(2, 16, 1, 1), (3, 37, 10, 10), (4, 300, 10, 4), (2, 1000, 3, 7), (1, 257, 64, 64)])
def test_normal_logweight_with_tensor_scales_is_bitwise_the_unfused_route(hip_device, dtype, B, K, dx, dy):
    """Learned scales — a per-dimension vector, a per-sequence [B,D] tensor, a full [B,K,D] network
    output — take K5's general kernel: log-weights bit-identical to K4 x 3 + K1, gradients (incl.
    those of the scales, through K4's backward) equal to the unfused route's."""
    from aesmc_amd import _ops
    gen = torch.Generator(device=hip_device).manual_seed(B * K + dx + dy)
    rand = lambda *shape: torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)
    pos = lambda *shape: torch.rand(*shape, device=hip_device, dtype=dtype, generator=gen) + 0.3
    weights = rand(B, K)
    base = dict(x=rand(B, K, dx), loc_p=rand(B, K, dx), y=rand(B, dy), loc_g=rand(B, K, dy), loc_q=rand(B, K, dx),
                s_p=pos(dx), s_g=pos(B, dy), s_q=pos(B, K, dx))

    def run(fused):
        t = {name: v.clone().requires_grad_() for name, v in base.items()}
        y = t["y"].unsqueeze(1).expand(B, K, dy)
        sp, sg, sq = t["s_p"].expand(B, K, dx), t["s_g"].unsqueeze(1).expand(B, K, dy), t["s_q"]
        if fused:
            lw = _ops.normal_log_weight(t["x"], t["loc_p"], sp, y, t["loc_g"], sg, t["loc_q"], sq)
            assert lw is not None
        else:
            lw = _ops.logweight_lse(_ops.normal_log_prob_sum(t["x"], t["loc_p"], sp),
                                    _ops.normal_log_prob_sum(y, t["loc_g"], sg),
                                    _ops.normal_log_prob_sum(t["x"], t["loc_q"], sq))[0]
        (lw * weights).sum().backward()
        return lw.detach(), {name: v.grad for name, v in t.items()}

    fused, unfused = run(True), run(False)
    assert torch.equal(fused[0], unfused[0])
    rtol = 1e-5 if dtype == torch.float32 else 1e-12
    for name in base:
        torch.testing.assert_close(fused[1][name], unfused[1][name], rtol=rtol, atol=rtol, msg=name)
    eager = (torch.distributions.Normal(base["loc_p"], base["s_p"]).log_prob(base["x"]).sum(-1)
             + torch.distributions.Normal(base["loc_g"], base["s_g"].unsqueeze(1)).log_prob(base["y"].unsqueeze(1)).sum(-1)
             - torch.distributions.Normal(base["loc_q"], base["s_q"]).log_prob(base["x"]).sum(-1))
    tol_ = 2e-5 if dtype == torch.float32 else 1e-12
    torch.testing.assert_close(fused[0], eager, rtol=tol_, atol=tol_ * (dx + dy))


def test_normal_logweight_declines_what_it_does_not_cover(kernels, hip_device):
    B, K = 2, 8
    x = torch.randn(B, K, 3, device=hip_device)
    y = torch.randn(B, K, 3, device=hip_device)
    one = torch.ones((), device=hip_device).expand(B, K, 3)
    vector_scale = torch.ones(3, device=hip_device).expand(B, K, 3)
    assert kernels.normal_logweight(x, x, vector_scale, y, y, one, x, one) is not None   # general kernel
    wide = torch.randn(B, K, 65, device=hip_device)
    one_w = torch.ones((), device=hip_device).expand(B, K, 65)
    assert kernels.normal_logweight(wide, wide, one_w, y, y, one, wide, one_w) is None    # 65 values: not whole vectors
    w128 = torch.randn(B, K, 128, device=hip_device)
    one128 = torch.ones((), device=hip_device).expand(B, K, 128)
    assert kernels.normal_logweight(w128, w128, one128, y, y, one, w128, one128) is None  # wide x, narrow y
    off = torch.randn(B, K, 129, device=hip_device)[:, :, 1:]                             # misaligned rows
    assert kernels.normal_logweight(w128, off, one128, w128, w128, one128, w128, one128) is None
    assert kernels.normal_logweight_covers(x, vector_scale, y, one, one)
    assert kernels.normal_logweight_covers(x, one, y, one, one)
    wide_vector = torch.ones(128, device=hip_device).expand(B, K, 128)
    assert kernels.normal_logweight(w128, w128, wide_vector, w128, w128, one128, w128, one128) is None  # wide + tensor scale


def test_normal_logweight_gradients_match_the_unfused_route(hip_device):
    from aesmc_amd import _ops
    B, K, d = 3, 50, 4
    gen = torch.Generator(device=hip_device).manual_seed(0)
    base = [torch.randn(B, K, d, device=hip_device, dtype=torch.float64, generator=gen) for _ in range(5)]
    log_scales = torch.randn(3, device=hip_device, dtype=torch.float64, generator=gen) * 0.1
    weights = torch.randn(B, K, device=hip_device, dtype=torch.float64, generator=gen)

    def run(fused):
        x, loc_p, y, loc_g, loc_q = [t.clone().requires_grad_() for t in base]
        ls = log_scales.clone().requires_grad_()
        sp, sg, sq = [ls[i].exp().expand(B, K, d) for i in range(3)]
        if fused:
            lw = _ops.normal_log_weight(x, loc_p, sp, y, loc_g, sg, loc_q, sq)
        else:
            lw = _ops.logweight_lse(_ops.normal_log_prob_sum(x, loc_p, sp), _ops.normal_log_prob_sum(y, loc_g, sg),
                                    _ops.normal_log_prob_sum(x, loc_q, sq))[0]
        (lw * weights).sum().backward()
        return lw.detach(), [t.grad for t in (x, loc_p, y, loc_g, loc_q, ls)]

    fused, unfused = run(True), run(False)
    assert torch.equal(fused[0], unfused[0])
    for a, b in zip(fused[1], unfused[1]):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)


def test_full_size_normal_logprob_sum(kernels, hip_device):
    """North-star shape: B=1024, K=4096, d=10."""
    B, K, d = 1024, 4096, 10
    gen = torch.Generator(device=hip_device).manual_seed(3)
    value = torch.randn(B, K, d, device=hip_device, generator=gen)
    loc = torch.randn(B, K, d, device=hip_device, generator=gen)
    scale = torch.tensor(0.7, device=hip_device)
    got = kernels.normal_logprob_sum(value, loc, scale.expand(value.shape))
    want = torch.distributions.Normal(loc, scale).log_prob(value).sum(-1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-4)
    # linearity in the scale: log N(v; mu, s) summed == -d log s + log N((v - mu)/s; 0, 1) summed
    unit = kernels.normal_logprob_sum((value - loc) / 0.7, torch.zeros_like(loc), torch.ones_like(loc))
    torch.testing.assert_close(got, unit - d * float(np.log(0.7)), rtol=1e-5, atol=1e-4)


# ---- K6 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 16, 1), (3, 7, 10), (5, 64, 3), (4, 1000, 10), (2, 33, 128),
                                   (3, 5, 2, 3)])
@pytest.mark.parametrize("loc_kind", ["full", "per_batch", "vector", "scalar"])
@pytest.mark.parametrize("scale_kind", ["scalar", "vector", "full"])
def test_normal_rsample_is_bitwise_eager_torch(kernels, hip_device, dtype, shape, loc_kind, scale_kind):
    gen = torch.Generator().manual_seed(sum(shape) * 131 + len(loc_kind) * 17 + len(scale_kind))
    make = lambda *s: torch.randn(tuple(s), generator=gen, dtype=dtype).to(hip_device)
    kinds = {"full": shape, "per_batch": (shape[0], 1) + shape[2:], "vector": shape[2:], "scalar": ()}
    eps = make(*shape)
    loc = make(*kinds[loc_kind])
    scale = make(*kinds[scale_kind]).abs() + 0.1
    got = kernels.normal_rsample(eps, loc.expand(shape), scale.expand(shape))
    want = loc + eps * scale
    assert got.shape == want.shape and got.is_contiguous()
    assert torch.equal(got, want)
    np.testing.assert_array_equal(got.cpu().numpy(),
                                  kernel_oracle.normal_rsample(eps.cpu().numpy(), loc.cpu().numpy(),
                                                               scale.cpu().numpy()))


def test_sample_draws_what_torch_draws_and_differentiates(hip_device):
    """state.sample through K6 == Distribution.rsample from the same generator state, for each
    BatchShapeMode; gradients w.r.t. loc and scale equal eager autograd's."""
    from aesmc_amd import state
    B, K, D = 4, 9, 5
    for dtype in (torch.float32, torch.float64):
        cases = [
            (lambda l, s: torch.distributions.Normal(l, s), (B, K, D), state.BatchShapeMode.FULLY_EXPANDED),
            (lambda l, s: torch.distributions.Independent(torch.distributions.Normal(l, s), 1), (B, K, D),
             state.BatchShapeMode.FULLY_EXPANDED),
            (lambda l, s: torch.distributions.Normal(l, s), (B, D), state.BatchShapeMode.BATCH_EXPANDED),
            (lambda l, s: torch.distributions.Normal(l, s), (D,), state.BatchShapeMode.NOT_EXPANDED),
        ]
        for make, loc_shape, mode in cases:
            loc = torch.randn(*loc_shape, dtype=dtype, device=hip_device, requires_grad=True)
            scale = (torch.rand(D, dtype=dtype, device=hip_device) + 0.5).requires_grad_()
            results = []
            for fused in (True, False):
                state.set_fused_normal(fused)
                try:
                    torch.manual_seed(11)
                    dist = state.set_batch_shape_mode(make(loc, scale.expand(loc_shape)), mode)
                    draw = state.sample(dist, B, K)
                    weights = torch.arange(draw.numel(), dtype=dtype, device=hip_device).reshape(draw.shape)
                    g_loc, g_scale = torch.autograd.grad((draw * weights.sin()).sum(), (loc, scale))
                finally:
                    state.set_fused_normal(True)
                results.append((draw.detach(), g_loc, g_scale))
            (d1, gl1, gs1), (d0, gl0, gs0) = results
            assert d1.shape == (B, K, D) and torch.equal(d1, d0)
            torch.testing.assert_close(gl1, gl0, rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(gs1, gs0, rtol=1e-5, atol=1e-5)


# ---- fused resampling step -----------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (2, 16), (3, 7), (5, 64), (4, 1000), (7, 1024), (3, 1025), (2, 4096),
                                   (2, 5000), (1, 20000), (1, 32768)])
@pytest.mark.parametrize("row", [None, (1,), (4,), (10,), (3, 4), (5,)])
def test_resample_step_equals_its_parts(kernels, hip_device, dtype, shape, row):
    """idx bit-exact == K2 (and the oracle), payload bit-exact == K3 on that idx, lse within the K1 bar."""
    B, K = shape
    rng = np.random.RandomState(B * 7919 + K + (0 if row is None else 13 * sum(row)))
    lw = dev((rng.randn(B, K) * 2.5).astype(dtype), hip_device)
    u = dev(rng.uniform(size=B), hip_device)
    payload = None if row is None else dev(rng.randn(B, K, *row).astype(np.float32), hip_device)
    out = kernels.resample_step(lw, u, payload, want_lse=True)
    want_idx = kernels.ancestor_index(lw, u)
    if out is None:  # declined: payload rows the fused launch does not cover
        assert payload is not None and (K * payload[0, 0].numel() * 4) % 16 != 0
        return
    idx, lse, moved = out
    assert torch.equal(idx, want_idx)
    np.testing.assert_array_equal(idx.cpu().numpy(),
                                  kernel_oracle.ancestor_index(lw.cpu().numpy(), u.cpu().numpy())[0])
    _, want_lse = kernel_oracle.logweight_lse(lw.cpu().numpy(), None, None)
    rtol, atol = tol(dtype)
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, rtol=rtol, atol=atol)
    if payload is None:
        assert moved is None
    else:
        assert torch.equal(moved, kernels.gather(payload, want_idx))
    assert kernels.read_flags(hip_device) == 0


def test_resample_step_strided_payload_and_special_rows(kernels, hip_device):
    from aesmc_amd import _lib
    rng = np.random.RandomState(5)
    B, K = 6, 128
    lw = rng.randn(B, K).astype(np.float32)
    lw[1, :] = -np.inf          # degenerate: index K, lse -inf
    lw[2, 5] = np.inf           # degenerate: lse +inf
    lw[3, 7] = np.nan           # NaN: lse NaN
    lw[4, ::2] = -np.inf        # fine: zero-weight particles
    lw_d, u = dev(lw, hip_device), dev(rng.uniform(size=B), hip_device)
    base = dev(rng.randn(B, 2 * K, 12).astype(np.float32), hip_device)
    payload = base[:, ::2, 2:10]                                    # strided in k, offset rows of 8 floats
    idx, lse, moved = kernels.resample_step(lw_d, u, payload, want_lse=True)
    flags = kernels.read_flags(hip_device)
    assert flags & _lib.FLAG_NAN_LOG_WEIGHT and flags & _lib.FLAG_DEGENERATE_ROW
    want_idx = kernels.ancestor_index(lw_d, u)
    assert torch.equal(idx, want_idx)
    want_moved = kernels.gather(payload, want_idx)                  # K3 clamps index K to K - 1
    kernels.read_flags(hip_device)
    assert torch.equal(moved, want_moved)
    lse = lse.cpu().numpy()
    assert lse[1] == -np.inf and lse[2] == np.inf and np.isnan(lse[3])
    good = [0, 4, 5]
    _, want_lse = kernel_oracle.logweight_lse(lw[good], None, None)
    np.testing.assert_allclose(lse[good], want_lse, rtol=F32_RTOL, atol=F32_ATOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("K,B,rest", [(1, 1, ()), (16, 4, ()), (33, 70, ()), (257, 129, ()), (8192, 16, ()),
                                      (64, 64, ()), (128, 192, ()), (512, 64, (1,)),
                                      (33, 70, (3,)), (100, 65, (10,)), (40, 9, (4, 5)), (17, 5, (100,)),
                                      (9, 3, (200,))])
@pytest.mark.parametrize("loc_kind", ["per_batch", "vector", "scalar"])
def test_normal_rsample_transposed_noise(kernels, hip_device, dtype, K, B, rest, loc_kind):
    """BATCH_EXPANDED layout: noise drawn as [K,B,...], result wanted as [B,K,...] (state.py:102-103)."""
    gen = torch.Generator().manual_seed(K * 31 + B)
    make = lambda *s: torch.randn(tuple(s), generator=gen, dtype=dtype).to(hip_device)
    shape = (K, B) + rest
    eps = make(*shape)
    loc = make(*{"per_batch": (B,) + rest, "vector": rest, "scalar": ()}[loc_kind])
    scale = make(*rest).abs() + 0.1
    got = kernels.normal_rsample(eps.transpose(0, 1), loc.expand(shape).transpose(0, 1),
                                 scale.expand(shape).transpose(0, 1))
    want = (loc + eps * scale).transpose(0, 1)
    assert got.shape == want.shape and got.is_contiguous()
    assert torch.equal(got, want)


# ---- K7 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1, ()), (2, 16, ()), (3, 7, (1,)), (5, 64, (3,)), (4, 1000, (10,)),
                                   (2, 33, (128,)), (3, 5, (2, 3)), (2, 300, (300,)), (1, 5000, (17,)),
                                   (2, 2000, (128,)), (3, 600, (20,)), (1, 700, (1024,)), (1, 300, (2048,)),
                                   (2, 257, (18,))])
@pytest.mark.parametrize("scale", [1.0, 8.0])
def test_particle_summary_matches_oracle(kernels, hip_device, dtype, shape, scale):
    B, K, tail = shape
    rng = np.random.RandomState(B * 131 + K)
    lw = (rng.randn(B, K) * scale).astype(dtype)
    value = rng.randn(B, K, *tail).astype(dtype)
    log_ess, mean, second = kernels.particle_summary(dev(lw, hip_device), dev(value, hip_device),
                                                     want_log_ess=True, want_mean=True, want_second=True)
    want_ess, want_mean, want_second = kernel_oracle.particle_summary(lw, value)
    rtol, atol = (2e-5, 2e-5) if dtype == np.float32 else (1e-12, 1e-12)
    assert mean.shape == (B,) + tail and second.shape == (B,) + tail and log_ess.shape == (B,)
    np.testing.assert_allclose(log_ess.cpu().numpy(), want_ess, rtol=rtol, atol=atol)
    np.testing.assert_allclose(mean.cpu().numpy(), want_mean, rtol=rtol, atol=atol)
    np.testing.assert_allclose(second.cpu().numpy(), want_second, rtol=rtol, atol=atol * 4)


def test_particle_summary_strided_value_offsets_and_broken_rows(kernels, hip_device):
    rng = np.random.RandomState(9)
    B, K = 5, 200
    base = rng.randn(B, 2 * K, 9)
    lw = rng.randn(B, K)
    lw[0] += 1e6                 # the reference's own offset cases (test/test_statistics.py:71-93)
    lw[1] -= 1e6
    lw[2, 3] = np.nan
    lw[3, :] = -np.inf
    lw[4, ::2] = -np.inf         # zero-weight particles are fine
    value = dev(base, hip_device)[:, ::2, 1:8]
    log_ess, mean, second = kernels.particle_summary(dev(lw, hip_device), value, want_log_ess=True,
                                                     want_mean=True, want_second=True)
    want = kernel_oracle.particle_summary(lw, base[:, ::2, 1:8])
    for got, expected in zip((log_ess, mean, second), want):
        np.testing.assert_allclose(got.cpu().numpy(), expected, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert np.isnan(log_ess.cpu().numpy()[[2, 3]]).all() and np.isnan(mean.cpu().numpy()[[2, 3]]).all()


def test_statistics_known_answers_of_the_reference(hip_device):
    """test/test_statistics.py:71-115 (log_ess / ess of [0.2, 0.3, 0.5] under offsets 0.47x, +1e6,
    -1e6) and the weighted mean / variance against their definitions, through aesmc_amd.statistics."""
    from aesmc_amd import statistics
    normalized = np.array([0.2, 0.3, 0.5])
    for log_weight in (np.log(normalized * 0.47), np.log(normalized) + 1e6, np.log(normalized) - 1e6):
        lw = torch.from_numpy(log_weight).to(hip_device)
        assert statistics.log_ess(lw).shape == ()
        np.testing.assert_allclose(statistics.log_ess(lw).item(), np.log(1 / np.sum(normalized ** 2)), atol=1e-7)
        np.testing.assert_allclose(statistics.ess(lw).item(), 1 / np.sum(normalized ** 2), atol=1e-7)
    assert statistics.log_ess(-torch.rand(3, 4, device=hip_device)).shape == (3,)
    gen = torch.Generator().manual_seed(3)
    value = torch.randn(6, 50, 4, generator=gen, dtype=torch.float64).to(hip_device)
    lw = torch.randn(6, 50, generator=gen, dtype=torch.float64).to(hip_device)
    w = torch.softmax(lw, dim=1).unsqueeze(-1)
    mean = (w * value).sum(1)
    torch.testing.assert_close(statistics.empirical_mean(value, lw), mean)
    torch.testing.assert_close(statistics.empirical_variance(value, lw), (w * value ** 2).sum(1) - mean ** 2)
    torch.testing.assert_close(statistics.empirical_expectation(value, lw, lambda x: x ** 2), (w * value ** 2).sum(1))
    # with gradients wanted the PyTorch expression runs and differentiates
    lw_g = lw.clone().requires_grad_()
    statistics.empirical_mean(value, lw_g).sum().backward()
    assert lw_g.grad is not None and torch.isfinite(lw_g.grad).all()


# ---- second, independent oracle: plain C (oracle/smc_core.c) at full sizes -------------------------
@pytest.mark.parametrize("B,K,d", [(256, 1024, 10), (1024, 4096, 10), (4, 40000, 3), (7, 333, 1)])
def test_hip_equals_c_oracle_at_full_size(kernels, hip_device, B, K, d):
    from oracle import c_oracle
    from aesmc_amd import inference
    rng = np.random.RandomState(B + K + d)
    lw = (rng.randn(B, K) * 2).astype(np.float32)
    lw[rng.randint(B), rng.randint(K)] = -np.inf
    u = rng.uniform(size=B)
    want_idx, flags = c_oracle.ancestor_index(lw, u)
    assert flags == 0
    idx = kernels.ancestor_index(dev(lw, hip_device), dev(u, hip_device))
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
    value = rng.randn(B, K, d).astype(np.float32)
    moved = kernels.gather(dev(value, hip_device), idx)
    np.testing.assert_array_equal(moved.cpu().numpy(), c_oracle.gather(value, want_idx)[0])
    grad = rng.randn(B, K, d).astype(np.float32)
    back = kernels.gather_backward(dev(grad, hip_device), idx, sorted_index=True)
    # float32 segmented sums against float64 ones: runs of a few hundred N(0,1) terms
    np.testing.assert_allclose(back.cpu().numpy(), c_oracle.gather_backward(grad, want_idx)[0], rtol=1e-4, atol=2e-4)
    # genealogy over three resampling steps (inference.get_resampled_latents vs the C lineage)
    steps = [c_oracle.ancestor_index((rng.randn(B, K) * 2).astype(np.float32), rng.uniform(size=B))[0] for _ in range(3)]
    latents = [rng.randn(B, K, d).astype(np.float32) for _ in range(4)]
    got = inference.get_resampled_latents([dev(x, hip_device) for x in latents], [dev(i, hip_device) for i in steps])
    for x, line, have in zip(latents, c_oracle.lineage(steps), got):
        np.testing.assert_array_equal(have.cpu().numpy(), c_oracle.gather(x, line)[0])
    assert kernels.read_flags(hip_device) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,K,dx,dy", [(3, 50, 4, 4), (2, 16, 1, 1), (2, 257, 10, 3), (1, 33, 128, 128)])
def test_normal_logweight_backward_is_bitwise_the_three_launch_route(hip_device, dtype, B, K, dx, dy):
    """Scales without gradients (buffers): K5's backward is ONE launch and must give, bit for bit, what
    three K4 backward launches and the eager add of the two x-gradients give — also for broadcast
    (BATCH_EXPANDED / NOT_EXPANDED) locations, whose gradients autograd's expand-backward reduces."""
    from aesmc_amd import _ops
    gen = torch.Generator(device=hip_device).manual_seed(B * K + dx)
    rand = lambda *shape: torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)
    scales = [torch.tensor(v, device=hip_device, dtype=dtype) for v in (1.0, 0.5, 0.7)]
    weights = rand(B, K)
    layouts = [
        dict(x=rand(B, K, dx), loc_p=rand(B, K, dx), y=rand(B, dy), loc_g=rand(B, K, dy), loc_q=rand(B, K, dx)),
        dict(x=rand(B, K, dx), loc_p=rand(dx), y=rand(B, dy), loc_g=rand(B, K, dy), loc_q=rand(B, dx)),
    ]
    for lay in layouts:
        def run(fused):
            leaves = {name: t.clone().requires_grad_() for name, t in lay.items()}
            x, y = leaves["x"], leaves["y"].unsqueeze(1).expand(B, K, dy)
            loc_p = leaves["loc_p"].expand(B, K, dx)
            loc_q = leaves["loc_q"].unsqueeze(1).expand(B, K, dx) if leaves["loc_q"].dim() == 2 else leaves["loc_q"]
            sp, sg, sq = scales[0].expand(B, K, dx), scales[1].expand(B, K, dy), scales[2].expand(B, K, dx)
            if fused:
                lw = _ops.normal_log_weight(x, loc_p, sp, y, leaves["loc_g"], sg, loc_q, sq)
            else:
                lw = _ops.logweight_lse(_ops.normal_log_prob_sum(x, loc_p, sp),
                                        _ops.normal_log_prob_sum(y, leaves["loc_g"], sg),
                                        _ops.normal_log_prob_sum(x, loc_q, sq))[0]
            (lw * weights).sum().backward()
            return lw.detach(), {name: t.grad for name, t in leaves.items()}
        fused, unfused = run(True), run(False)
        assert torch.equal(fused[0], unfused[0])
        for name in lay:
            assert torch.equal(fused[1][name], unfused[1][name]), name


def test_fused_normal_ops_on_random_views_match_eager_torch(hip_device):
    """Fuzz: state.log_prob (K4), state.normal_log_weight (K5) and state.sample (K6) on randomly
    shaped, sliced, transposed and broadcast operands against the eager PyTorch expressions
    (set_fused_normal(False)), values and gradients."""
    from aesmc_amd import state
    rng = np.random.RandomState(2024)
    gen = torch.Generator(device=hip_device).manual_seed(7)
    Normal = torch.distributions.Normal

    def random_view(shape, dtype):
        """A tensor of `shape` that may be a slice / transpose of a larger buffer (never a copy)."""
        kind = rng.randint(4)
        full = list(shape)
        if kind == 1 and len(shape) >= 2:          # every other particle
            full[1] *= 2
            return torch.randn(*full, device=hip_device, dtype=dtype, generator=gen)[:, ::2]
        if kind == 2 and len(shape) >= 3:          # a window of a wider last dim
            full[-1] += 3
            return torch.randn(*full, device=hip_device, dtype=dtype, generator=gen)[..., 1:1 + shape[-1]]
        if kind == 3 and len(shape) >= 2:          # transposed leading dims
            full[0], full[1] = full[1], full[0]
            return torch.randn(*full, device=hip_device, dtype=dtype, generator=gen).transpose(0, 1)
        return torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)

    for case in range(40):
        dtype = [torch.float32, torch.float64][case % 2]
        B, K = int(rng.randint(1, 6)), int(rng.randint(1, 70))
        tail = [(), (1,), (3,), (10,), (2, 3), (70,)][rng.randint(6)]
        shape = (B, K) + tail
        loc_shapes = [shape, (B,) + tail, tail]                       # FULLY / BATCH / NOT expanded
        modes = [state.BatchShapeMode.FULLY_EXPANDED, state.BatchShapeMode.BATCH_EXPANDED,
                 state.BatchShapeMode.NOT_EXPANDED]
        which = rng.randint(3, size=3)
        scale = [torch.tensor(float(rng.uniform(0.3, 2.0)), device=hip_device, dtype=dtype) for _ in range(3)]
        base = dict(x=random_view(shape, dtype), y=random_view((B,) + tail, dtype),
                    **{"loc%d" % i: random_view(loc_shapes[which[i]], dtype) if loc_shapes[which[i]] else
                       torch.randn((), device=hip_device, dtype=dtype, generator=gen) for i in range(3)})
        weights = torch.randn(B, K, device=hip_device, dtype=dtype, generator=gen)

        def run(fused):
            state.set_fused_normal(fused)
            try:
                leaves = {name: t.detach().clone().requires_grad_() if False else t.detach().requires_grad_()
                          for name, t in base.items()}
                dists = [state.set_batch_shape_mode(Normal(leaves["loc%d" % i], scale[i], validate_args=False),
                                                    modes[which[i]]) for i in range(3)]
                obs = state.expand_observation(leaves["y"], K)
                lw = None
                if fused:
                    lw = state.normal_log_weight(dists[0], dists[2], leaves["x"], dists[1], obs)
                if lw is None:
                    lw = state.log_prob(dists[0], leaves["x"]) + state.log_prob(dists[1], obs) \
                        - state.log_prob(dists[2], leaves["x"])
                torch.manual_seed(case)
                draw = state.sample(dists[2], B, K)
                total = (lw * weights).sum() + (draw * draw.detach().cos()).sum()
                grads = torch.autograd.grad(total, list(leaves.values()), allow_unused=True)
                return lw.detach(), draw.detach(), grads
            finally:
                state.set_fused_normal(True)

        fused, eager = run(True), run(False)
        rtol, atol = (2e-5, 2e-4) if dtype == torch.float32 else (1e-11, 1e-10)
        torch.testing.assert_close(fused[0], eager[0], rtol=rtol, atol=atol, msg=str((case, shape, which)))
        assert torch.equal(fused[1], eager[1]), (case, shape, which)           # the draw: bit for bit
        for name, a, b in zip(base, fused[2], eager[2]):
            if a is None or b is None:
                assert a is None and b is None, (case, name)
                continue
            torch.testing.assert_close(a, b, rtol=rtol * 10, atol=atol * 10, msg=str((case, name, shape, which)))


def test_many_short_rows(kernels, hip_device):
    """A very large batch of very short particle systems (B = 200 000, K = 3): every kernel's grid
    is sized by B here; indices and rows against the C oracle, sums against float64."""
    from oracle import c_oracle
    B, K, d = 200000, 3, 2
    rng = np.random.RandomState(12)
    lw = rng.randn(B, K).astype(np.float32)
    u = rng.uniform(size=B)
    x = rng.randn(B, K, d).astype(np.float32)
    idx, lse, moved = kernels.resample_step(dev(lw, hip_device), dev(u, hip_device), dev(x, hip_device), want_lse=True) \
        or (None, None, None)
    want_idx, _ = c_oracle.ancestor_index(lw, u)
    if idx is None:      # K * row_bytes = 24 is not a multiple of 16: the fused payload is declined
        idx = kernels.ancestor_index(dev(lw, hip_device), dev(u, hip_device))
        moved = kernels.gather(dev(x, hip_device), idx)
        _, lse = kernels.logweight_lse(dev(lw, hip_device), None, None, want_lw=False, want_lse=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
    np.testing.assert_array_equal(moved.cpu().numpy(), c_oracle.gather(x, want_idx)[0])
    np.testing.assert_allclose(lse.cpu().numpy(), c_oracle.logweight_lse(lw)[1], rtol=2e-6, atol=2e-6)
    back = kernels.gather_backward(dev(x, hip_device), idx, sorted_index=True)
    np.testing.assert_allclose(back.cpu().numpy(), c_oracle.gather_backward(x, want_idx)[0], rtol=1e-5, atol=1e-5)
    scale = torch.tensor(0.7, device=hip_device).expand(B, K, d)
    lp = kernels.normal_logprob_sum(dev(x, hip_device), dev(x[:, ::-1].copy(), hip_device), scale)
    want = torch.distributions.Normal(torch.from_numpy(x[:, ::-1].copy()).double(), 0.7).log_prob(
        torch.from_numpy(x).double()).sum(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    ess, mean, _ = kernels.particle_summary(dev(lw, hip_device), dev(x, hip_device), True, True, False)
    want_ess, want_mean, _ = kernel_oracle.particle_summary(lw, x)
    np.testing.assert_allclose(ess.cpu().numpy(), want_ess, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(mean.cpu().numpy(), want_mean, rtol=2e-5, atol=2e-5)
    assert kernels.read_flags(hip_device) == 0


# ---- round 2: workgroups sharing a row, the range backward, K1's running sum ------------------------
@pytest.mark.parametrize("shape,row", [((2, 64), (4,)), ((3, 1024), (10,)), ((5, 4096), (10,)), ((2, 2048), (3, 4)),
                                       ((4, 600), (4,)), ((1, 16384), (2,)), ((3, 1000), (1,))])
def test_resample_step_does_not_depend_on_workgroups_per_row(kernels, hip_device, shape, row):
    """`aesmc_test_set_step_parts`: a batch row shared by 1, 2, 4 or 8 workgroups gives the same indices,
    log-sum-exp and payload bit for bit — including degenerate and NaN rows."""
    B, K = shape
    rng = np.random.RandomState(K + B)
    lw = (rng.randn(B, K) * 2).astype(np.float32)
    if B >= 3:
        lw[1, :] = -np.inf
        lw[2, K // 2] = np.nan
    lw_d, u = dev(lw, hip_device), dev(rng.uniform(size=B), hip_device)
    payload = dev(rng.randn(B, K, *row).astype(np.float32), hip_device)
    lib = kernels._lib
    try:
        assert lib.aesmc_test_set_step_parts(3) != 0          # powers of two only
        results = []
        for parts in (1, 2, 4, 8, 0):
            assert lib.aesmc_test_set_step_parts(parts) == 0
            out = kernels.resample_step(lw_d, u, payload, want_lse=True)
            assert out is not None
            results.append(out)
    finally:
        lib.aesmc_test_set_step_parts(0)
    kernels.read_flags(hip_device)
    for idx, lse, moved in results[1:]:
        assert torch.equal(idx, results[0][0]) and torch.equal(moved, results[0][2])
        np.testing.assert_array_equal(lse.cpu().numpy(), results[0][1].cpu().numpy())
    good = [b for b in range(B) if np.isfinite(lw[b]).any() and not np.isnan(lw[b]).any()]
    want, _ = kernel_oracle.ancestor_index(lw[good], u.cpu().numpy()[good])
    np.testing.assert_array_equal(results[0][0].cpu().numpy()[good], want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,scale", [((2, 16), 1.0), ((4, 100, 10), 1.0), ((2, 300, 10), 8.0), ((2, 1000, 6), 30.0),
                                         ((1, 5000, 10), 1.0), ((1, 5000, 10), 40.0), ((3, 700, 128), 3.0),
                                         ((2, 2049, 1), 2.0), ((300, 33, 4), 1.0), ((2, 600, 129), 0.0),
                                         ((5, 1024, 16), 200.0), ((2, 4096, 10), 0.0), ((2, 4096, 10), 3.0)])
def test_sorted_backward_kernels_agree_bit_for_bit(kernels, hip_device, dtype, shape, scale):
    """The range kernel (every row of the gradient written once, no zero fill) and round 1's kernel
    behind a zero fill sum the same rows in the same order when their tiles coincide (rows up to
    112 bytes: 256 particles per tile in both): identical bits, also into a destination that held
    garbage (nothing may rely on a previous fill).  Wider rows tile differently (28 vs 32 KiB of
    staging), so a run crossing a tile boundary is associated differently: equal to rounding."""
    rng = np.random.RandomState(shape[1] + int(scale) + 1)
    go = dev(rng.randn(*shape).astype(dtype), hip_device)
    idx = dev(sorted_indices(rng, shape[0], shape[1], scale), hip_device)
    lib = kernels._lib
    try:
        outs = []
        for which in (1, 0):
            assert lib.aesmc_test_set_sorted_backward_kernel(which) == 0
            torch.full(shape, float("nan"), dtype=go.dtype, device=hip_device)   # dirty the allocator's blocks
            outs.append(kernels.gather_backward(go, idx, sorted_index=True))
    finally:
        lib.aesmc_test_set_sorted_backward_kernel(0)
    assert kernels.read_flags(hip_device) == 0
    row_bytes = int(np.prod(shape[2:])) * go.element_size()
    if row_bytes <= 112:
        assert torch.equal(outs[0], outs[1])
    else:
        rtol, atol = tol(dtype)
        torch.testing.assert_close(outs[0], outs[1], rtol=50 * rtol, atol=200 * atol)
    assert bool(torch.isfinite(outs[1]).all())


def test_sorted_backward_range_kernel_edge_rows(kernels, hip_device):
    """Ranges that start or end a batch row, runs spanning many tiles, a single survivor, the identity,
    an out-of-range entry and a descent (both flagged, never fatal)."""
    K, d = 1300, 3
    rng = np.random.RandomState(1)
    go = rng.randn(7, K, d)
    idx = np.stack([np.zeros(K, np.int64), np.full(K, K - 1, np.int64), np.arange(K, dtype=np.int64),
                    np.sort(rng.randint(0, K, size=K)).astype(np.int64),
                    np.sort(rng.randint(K - 5, K, size=K)).astype(np.int64),
                    np.sort(rng.randint(0, 3, size=K)).astype(np.int64),
                    np.repeat(np.arange(0, K, 260), 260)[:K].astype(np.int64)])
    got = kernels.gather_backward(dev(go, hip_device), dev(idx, hip_device), sorted_index=True).cpu().numpy()
    want, _ = kernel_oracle.gather_backward(go, idx)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-10)
    assert kernels.read_flags(hip_device) == 0
    bad = idx.copy()
    bad[3, K - 1] = K + 7                                  # beyond the row: contributes nothing
    got = kernels.gather_backward(dev(go, hip_device), dev(bad, hip_device), sorted_index=True).cpu().numpy()
    assert kernels.read_flags(hip_device) & kernel_oracle.FLAG_INDEX_OUT_OF_RANGE
    want, _ = kernel_oracle.gather_backward(go, bad)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-10)
    bad = idx.copy()
    bad[3, 700] = 0
    kernels.gather_backward(dev(go, hip_device), dev(bad, hip_device), sorted_index=True)
    assert kernels.read_flags(hip_device) & 16


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (3, 7), (5, 64), (4, 1000), (7, 1024), (3, 1025), (2, 8192), (300, 33)])
def test_logweight_accumulate_matches_oracle(kernels, hip_device, dtype, shape):
    """K1 with the running sum of importance sampling: every sum is one IEEE add in the oracle's
    order, so log-weights and totals are exact; the log-sum-exp of the total within K1's bar."""
    rng = np.random.RandomState(shape[1])
    a, b, c, acc = [(3 * rng.randn(*shape)).astype(dtype) for _ in range(4)]
    args = [dev(x, hip_device) for x in (a, b, c, acc)]
    lw, total, lse = kernels.logweight_accumulate(*args, want_lw=True, want_lse=True)
    want_lw, want_total, want_lse = kernel_oracle.logweight_accumulate(a, b, c, acc)
    np.testing.assert_array_equal(lw.cpu().numpy(), want_lw)
    np.testing.assert_array_equal(total.cpu().numpy(), want_total)
    rtol, atol = tol(dtype)
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, rtol=rtol, atol=atol)
    # the step's weights come from K5: a alone, nothing to combine
    lw, total, lse = kernels.logweight_accumulate(args[0], None, None, args[3], want_lw=False, want_lse=False)
    assert lw is None and lse is None
    np.testing.assert_array_equal(total.cpu().numpy(), (acc + a).astype(dtype))


def test_importance_sampling_running_sum_equals_the_stack_and_differentiates(hip_device):
    """infer('is') over T = 5 steps: log_weight equals the left-to-right sum of the per-step weights
    bit for bit (the order the reference's CPU torch.sum takes over the leading dim of its stack;
    the device's own torch.sum associates differently, hence only allclose against that), log Z its
    logsumexp, and the loss gradient equals autograd through the eager stack + sum + logsumexp."""
    from aesmc_amd import inference
    from aesmc_amd.testing import models, replay
    B, K, T, d = 3, 50, 5, 2
    for dtype in (torch.float64, torch.float32):
        model = models.LgssmNd(d, seed=1, dtype=dtype, validate_args=False).to(hip_device)
        observations = model.simulate(T, B, seed=2)
        torch.manual_seed(0)
        with replay.record() as tape:
            out = inference.infer("is", observations, model.initial, model.transition, model.emission, model.proposal,
                                  K, return_log_marginal_likelihood=True, return_log_weights=True, return_latents=False)
        stacked = torch.sum(torch.stack(out["log_weights"], dim=0), dim=0)
        running = out["log_weights"][0]
        for step_weights in out["log_weights"][1:]:
            running = running + step_weights
        assert torch.equal(out["log_weight"], running)
        torch.testing.assert_close(out["log_weight"], stacked, rtol=1e-6, atol=1e-5)
        want_lml = torch.logsumexp(stacked, dim=1) - np.log(K)
        torch.testing.assert_close(out["log_marginal_likelihood"], want_lml, rtol=1e-6, atol=1e-6)
        model.zero_grad()
        (-out["log_marginal_likelihood"].mean()).backward()
        mine = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        model.zero_grad()
        with replay.replay(tape):
            again = inference.infer("is", observations, model.initial, model.transition, model.emission,
                                    model.proposal, K, return_log_weights=True, return_latents=False,
                                    return_log_weight=False)
        eager = torch.logsumexp(torch.sum(torch.stack(again["log_weights"], dim=0), dim=0), dim=1) - np.log(K)
        (-eager.mean()).backward()
        theirs = [p.grad for p in model.parameters() if p.grad is not None]
        assert len(mine) == len(theirs) > 0
        for g, h in zip(mine, theirs):
            scale = float(h.abs().max()) + 1e-30
            torch.testing.assert_close(g / scale, h / scale, rtol=0, atol=1e-5 if dtype == torch.float32 else 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,K,dx,dy", [(3, 50, 4, 4), (2, 257, 10, 3), (1, 33, 128, 128), (40, 8192, 16, 16),
                                       (7, 1000, 10, 10), (5, 64, 1, 1), (2, 6, 3, 2), (3, 5, 3, 3), (64, 4096, 10, 10)])
def test_normal_logweight_backward_dense_kernel_equals_the_generic_one(kernels, hip_device, dtype, B, K, dx, dy):
    """The Markov-model layout (dense x and locations, the observation one row per batch element,
    scalar scales) takes `normal_logweight_bwd_dense_kernel`: 16-byte vectors, constant-step index
    arithmetic.  The same values behind a strided view take the generic kernel.  Identical bits for every
    gradient, with the gradient of the log-weights given and with K1's softmax term formed in place."""
    gen = torch.Generator(device=hip_device).manual_seed(B + K + dx)
    rand = lambda *shape: torch.randn(*shape, device=hip_device, dtype=dtype, generator=gen)
    x, loc_p, loc_q, loc_g = rand(B, K, dx), rand(B, K, dx), rand(B, K, dx), rand(B, K, dy)
    y = rand(B, dy).unsqueeze(1).expand(B, K, dy)
    sp, sg, sq = [torch.tensor(v, device=hip_device, dtype=dtype) for v in (1.0, 0.5, 0.7)]
    scales = (sp.expand(B, K, dx), sg.expand(B, K, dy), sq.expand(B, K, dx))
    padded = torch.zeros(B, K, dx + 1, device=hip_device, dtype=dtype)
    padded[:, :, :dx] = loc_p
    strided = padded[:, :, :dx]                          # same values, rows dx + 1 apart: not the dense layout
    assert torch.equal(strided, loc_p) and not strided.is_contiguous()
    grad_lw, grad_lse = rand(B, K), rand(B)
    lw = kernels.normal_logweight(x, loc_p, scales[0], y, loc_g, scales[1], loc_q, scales[2])
    _, lse = kernels.logweight_lse(lw, None, None, want_lw=False)
    need = [True, True, False, False, True, False, True, False]
    for extra in (dict(), dict(lw=lw, lse=lse, grad_lse=grad_lse)):
        given = None if extra else grad_lw
        dense = kernels.normal_logweight_backward(x, loc_p, scales[0], y, loc_g, scales[1], loc_q, scales[2], given,
                                                  need, **extra)
        generic = kernels.normal_logweight_backward(x, strided, scales[0], y, loc_g, scales[1], loc_q, scales[2],
                                                    given, need, **extra)
        for wanted, a, b in zip(need, dense, generic):
            assert (a is None) == (not wanted) and (b is None) == (not wanted)
            if wanted:
                assert torch.equal(a, b)
        assert bool(torch.isfinite(dense[0]).all())
    # and against K1's backward followed by the generic kernel (the two-launch route)
    g, _ = kernels.logweight_lse_backward(lw, lse, None, grad_lse, want_neg=False)
    two = kernels.normal_logweight_backward(x, strided, scales[0], y, loc_g, scales[1], loc_q, scales[2], g, need)
    one = kernels.normal_logweight_backward(x, loc_p, scales[0], y, loc_g, scales[1], loc_q, scales[2], None, need,
                                            lw=lw, lse=lse, grad_lse=grad_lse)
    for wanted, a, b in zip(need, one, two):
        if wanted:
            assert torch.equal(a, b)
