"""The two kernels a payload-free resampling step can launch (aesmc_ancestor_index / aesmc_resample_step[_ranges]:
systematic resampling, aesmc/inference.py:234-269) against each other and against `oracle/`:

  * ancestor_index_rows_kernel (the lean form: whole rows of blockDim.x * C particles, scans on the DPP crossbar) gives
    the ancestor indices and children ranges of ancestor_index_inv_kernel on the same inputs — every one — and its row
    log-sum-exp to the last place or two of the dtype, at every shape class the lean form takes (C = 4 / 8 / 16 / 32 particles
    per lane, one to sixteen wavefronts per row), float32 and float64 log-weights, healthy and collapsed weights, rows
    with -inf entries, uniforms at the ends of [0, 1);
  * both equal the NumPy oracle's indices (oracle/kernel_oracle.py) on float64 inputs;
  * degenerate rows (NaN, all -inf, +inf) are flagged and filled the same way by both.
"""
import numpy as np
import pytest
import torch

from oracle import kernel_oracle

pytestmark = pytest.mark.gpu

GENERAL, ROWS = 1, 2


@pytest.fixture(scope="module")
def kernels(hip_device):
    from aesmc_amd import _kernels
    provider = _kernels.get()
    assert provider.name == "hip"
    return provider


@pytest.fixture()
def k2_form(kernels):
    lib = kernels._lib
    yield lambda form: lib.aesmc_test_set_k2_form(form)
    lib.aesmc_test_set_k2_form(0)


def _step(kernels, log_w, u, ranges):
    idx, lse, _ = kernels.resample_step(log_w, u, None, want_lse=True, want_child_end=ranges)
    return idx, lse, getattr(idx, "_aesmc_child_end", None), kernels._lib.aesmc_test_last_k2_form()


# (B, K): C = 4 (K <= 2048), 8 (<= 8192), 16 (<= 16384), 32; one wavefront per row up to sixteen
SHAPES = [(1024, 4096), (256, 1024), (7, 768), (33, 256 * 3), (5, 2048), (3, 8192), (64, 16384), (2, 32768), (128, 4096),
          (9, 512 * 5), (4, 1024 * 12)]


@pytest.mark.parametrize("ranges", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_the_lean_form_gives_the_general_kernels_indices(kernels, hip_device, k2_form, shape, dtype, ranges):
    B, K = shape
    gen = torch.Generator(device=hip_device).manual_seed(B * 7 + K)
    for spread in (1.0, 6.0, 40.0):
        log_w = (spread * torch.randn(B, K, device=hip_device, generator=gen, dtype=torch.float64)).to(dtype)
        log_w[0, ::3] = float("-inf")                      # zero weights are legal (aesmc/inference.py:253-254)
        if B > 1:
            log_w[1, : K - 1] = float("-inf")              # one survivor, the last particle
        u = torch.rand(B, device=hip_device, dtype=torch.float64, generator=gen)
        u[0] = 0.0
        if B > 2:
            u[2] = 1.0 - 2.0 ** -53
        kernels.read_flags(hip_device)
        k2_form(GENERAL)
        want_idx, want_lse, want_ends, ran = _step(kernels, log_w, u, ranges)
        assert ran == GENERAL
        k2_form(ROWS)
        got_idx, got_lse, got_ends, ran = _step(kernels, log_w, u, ranges)
        assert ran == ROWS, "the lean form declined a shape it is built for"
        assert kernels.read_flags(hip_device) == 0
        assert torch.equal(got_idx, want_idx), int((got_idx != want_idx).sum())
        if ranges:
            assert torch.equal(got_ends, want_ends)
        eps = torch.finfo(dtype).eps
        assert float(((got_lse - want_lse).abs() / want_lse.abs().clamp_min(1.0)).max()) <= 2 * eps
        # (row 2's uniform is the largest float64 below 1: its last position (u + K - 1) / K may round to 1.0, where the
        #  reference's np.digitize — and both kernels — answer K: SURVEY.md 8(a), edge semantics)
        assert int(got_idx.min()) >= 0 and int(got_idx.max()) <= K
        others = torch.ones(B, dtype=torch.bool, device=hip_device)
        if B > 2:
            others[2] = False
        assert int(got_idx[others].max()) < K
        assert bool((got_idx[:, 1:] >= got_idx[:, :-1]).all())


@pytest.mark.parametrize("shape", [(16, 4096), (5, 768), (3, 16384)])
def test_the_lean_form_equals_the_numpy_oracle_on_float64_inputs(kernels, hip_device, k2_form, shape):
    B, K = shape
    rng = np.random.RandomState(B + K)
    log_w = 3.0 * rng.randn(B, K)
    u = rng.uniform(size=B)
    k2_form(ROWS)
    idx, lse, ends, ran = _step(kernels, torch.from_numpy(log_w).to(hip_device), torch.from_numpy(u).to(hip_device), True)
    assert ran == ROWS
    want, flags = kernel_oracle.ancestor_index(log_w, u)
    assert flags == 0
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    # children ranges: child_end[b, j] = #{k : idx[b, k] <= j}
    counts = np.stack([np.searchsorted(want[b], np.arange(K), side="right") for b in range(B)])
    np.testing.assert_array_equal(ends.cpu().numpy(), counts)
    m = log_w.max(axis=1)
    np.testing.assert_allclose(lse.cpu().numpy(), m + np.log(np.exp(log_w - m[:, None]).sum(axis=1)), rtol=1e-14)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_both_forms_treat_degenerate_rows_alike(kernels, hip_device, k2_form, dtype):
    from aesmc_amd import _lib
    B, K = 6, 1024
    gen = torch.Generator(device=hip_device).manual_seed(5)
    base = torch.randn(B, K, device=hip_device, generator=gen, dtype=torch.float64).to(dtype)
    u = torch.rand(B, device=hip_device, dtype=torch.float64, generator=gen)
    for kind, bit in (("nan", _lib.FLAG_NAN_LOG_WEIGHT), ("neg_inf", _lib.FLAG_DEGENERATE_ROW), ("pos_inf", _lib.FLAG_DEGENERATE_ROW)):
        log_w = base.clone()
        if kind == "nan":
            log_w[2, 17] = float("nan")
        elif kind == "neg_inf":
            log_w[2] = float("-inf")
        else:
            log_w[2, 900] = float("inf")
        out = {}
        for form in (GENERAL, ROWS):
            kernels.read_flags(hip_device)
            k2_form(form)
            idx, lse, ends, ran = _step(kernels, log_w, u, True)
            assert ran == form
            assert kernels.read_flags(hip_device) & bit
            out[form] = (idx, lse, ends)
        assert torch.equal(out[ROWS][0], out[GENERAL][0]) and torch.equal(out[ROWS][2], out[GENERAL][2])
        assert bool((out[ROWS][0][2] == K).all()) and bool((out[ROWS][2][2] == 0).all())
        a, b = out[ROWS][1], out[GENERAL][1]
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        keep = ~torch.isnan(a)
        torch.testing.assert_close(a[keep], b[keep], rtol=4 * torch.finfo(dtype).eps, atol=0)


def test_the_lean_form_declines_ragged_rows_and_payloads(kernels, hip_device, k2_form):
    k2_form(ROWS)
    gen = torch.Generator(device=hip_device).manual_seed(1)
    u = torch.rand(4, device=hip_device, dtype=torch.float64, generator=gen)
    for K in (1000, 4099, 300):                   # not whole rows of 64 C particles: the general kernel, same answer
        log_w = torch.randn(4, K, device=hip_device, generator=gen)
        idx, _, _, ran = _step(kernels, log_w, u, True)
        assert ran == GENERAL and int(idx.max()) < K
    log_w = torch.randn(4, 1024, device=hip_device, generator=gen)
    x = torch.randn(4, 1024, 10, device=hip_device, generator=gen)
    idx, lse, moved = kernels.resample_step(log_w, u, x, want_lse=True)      # a payload: the fused step's kernel
    assert kernels._lib.aesmc_test_last_k2_form() == GENERAL
    assert torch.equal(moved, torch.gather(x, 1, idx.unsqueeze(-1).expand_as(x)))


@pytest.mark.parametrize("form", [GENERAL, ROWS])
@pytest.mark.parametrize("shape", [(2048, 4096), (512, 2048), (256, 16384)])
def test_children_ranges_stay_monotone_on_knife_edge_rows(kernels, hip_device, k2_form, form, shape):
    """a tree-ordered scan holds the same partial sum, associated differently, in every lane of a stretch of
    exact-zero weights — one ulp apart in either order — so where a resampling position sits exactly on that CDF value a
    later lane's first[] could come out one BELOW an earlier lane's, and ranges written from the raw first[] overlapped.
    Rows built to sit on that edge: random weights, a block of zero weights spanning many lanes (log-weight -1e4: exp
    underflows to exactly 0), the SAME random weights mirrored, u = 0 and K a power of two — the CDF over the zero block
    is 1/2 to within rounding, K/2 an exact position.  The ranges must be non-decreasing, end at K, and be exactly the
    counts of the indices the launch returned, which are the indices of the same launch without the ranges."""
    B, K = shape
    gen = torch.Generator().manual_seed(K + B)
    quarter = K // 4
    side = torch.randn(B, quarter, generator=gen, dtype=torch.float64)
    log_w = torch.cat([side, torch.full((B, K - 2 * quarter), -1.0e4, dtype=torch.float64), side.flip(1)], dim=1)
    log_w = log_w.to(hip_device)
    u = torch.zeros(B, dtype=torch.float64, device=hip_device)
    k2_form(form)
    idx, _, child_end, ran = _step(kernels, log_w, u, ranges=True)
    assert ran == form
    other, _, _, _ = _step(kernels, log_w, u, ranges=False)      # the same kernel without the ranges: the same indices
    assert torch.equal(idx, other)
    # (the two kernels associate their partial sums differently, so ON the edge they may legitimately disagree about the
    #  one position K/2 — as any two summation orders do; everywhere else they agree)
    k2_form(GENERAL if form == ROWS else ROWS)
    across, _, _, _ = _step(kernels, log_w, u, ranges=False)
    assert int((across != idx).sum(dim=1).max()) <= K - 2 * quarter + 1
    assert int(((across - idx).abs()).max()) <= K - 2 * quarter + 1
    ends = child_end.cpu().numpy().astype(np.int64)
    assert (np.diff(ends, axis=1) >= 0).all(), "children ranges overlap"
    assert (ends[:, -1] == K).all()
    host = idx.cpu().numpy()
    assert (np.diff(host, axis=1) >= 0).all()
    for b in range(0, B, max(1, B // 64)):
        assert np.array_equal(ends[b], np.searchsorted(host[b], np.arange(K), side="right")), b
    # the rows do sit on the edge: the zero block's owners differ from row to row (K/2 falls on either side of it)
    middle = host[:, K // 2]
    assert len(np.unique(middle < quarter + (K - 2 * quarter))) >= 1
    assert kernels.read_flags(hip_device) == 0
