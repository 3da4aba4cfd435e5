"""ORACLE — TEST INFRASTRUCTURE ONLY.  NumPy restatement of the CONTRACT of each HIP kernel
(include/aesmc_hip.h): what K1 / K2 / K3 must return for given inputs, written independently of
the kernels' parallel structure.  The HIP results are compared against these; these in turn are
pinned to the reference by tests/golden/*.npz and to oracle/reference_port.py (the op-for-op port)
in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Where the contract deliberately departs from the reference's arithmetic (and why):
  * K2 evaluates the weight CDF in float64 whatever the input dtype.  The reference works in the
    input dtype (float32 CDF by sequential np.cumsum, aesmc/inference.py:253-261).  For float64
    inputs both agree exactly; for float32 inputs ~1e-4 of the indices differ by +-1 because the
    reference's float32 CDF carries ~1e-5 rounding noise (hard part 1).
    `ancestor_index_reference_dtype` below keeps the reference's dtype for comparison.
"""
import numpy as np

FLAG_NAN_LOG_WEIGHT = 1
FLAG_DEGENERATE_ROW = 2
FLAG_INDEX_OUT_OF_RANGE = 4


def logweight_lse(a, b=None, c=None):
    """K1.  lw = a + b - c in the input dtype (aesmc/inference.py:97-98, :125-126);
    lse = logsumexp over axis 1 with torch.logsumexp's special-value rules (inference.py:130)."""
    lw = np.array(a, copy=True)
    if b is not None:
        lw = lw + b
    if c is not None:
        lw = lw - c
    lw = lw.astype(a.dtype, copy=False)
    with np.errstate(all="ignore"):
        m = np.max(lw, axis=1, keepdims=True) if lw.shape[1] else np.full((lw.shape[0], 1), -np.inf, lw.dtype)
        shift = np.where(np.isfinite(m), m, 0).astype(lw.dtype)
        lse = (np.log(np.sum(np.exp(lw - shift), axis=1, keepdims=True, dtype=lw.dtype)) + shift)[:, 0]
    return lw, lse.astype(a.dtype)


def logweight_accumulate(a, b, c, acc):
    """K1 with the running sum of importance sampling (aesmc/inference.py:156-159: torch.sum over a
    stack of the per-step weights, left to right): lw = a + b - c, total = acc + lw in the input
    dtype, lse = logsumexp of total over axis 1.  Returns (lw, total, lse)."""
    lw, _ = logweight_lse(a, b, c)
    total = (acc + lw).astype(a.dtype, copy=False)
    _, lse = logweight_lse(total)
    return lw, total, lse


def logweight_lse_backward(lw, lse, grad_lw, grad_lse):
    """K1 backward: g = grad_lw + grad_lse * softmax(lw); returns (g, -g)."""
    g = np.zeros_like(lw)
    if grad_lse is not None:
        with np.errstate(all="ignore"):
            g = g + grad_lse[:, None] * np.exp(lw - lse[:, None])
    if grad_lw is not None:
        g = g + grad_lw
    g = g.astype(lw.dtype, copy=False)
    return g, -g


def ancestor_index(log_w, u):
    """K2 contract: float64 CDF, idx[b,k] = #{j : c[b,j] <= (u[b] + k) / K}  (the reference's
    np.digitize(pos, c), aesmc/inference.py:250-264).  Returns (idx int64 [B,K], flags)."""
    log_w = np.asarray(log_w)
    B, K = log_w.shape
    u = np.asarray(u, dtype=np.float64).reshape(B)
    idx = np.empty((B, K), dtype=np.int64)
    flags = 0
    pos_k = np.arange(0, K)
    for b in range(B):
        row = log_w[b]
        if np.isnan(row).any():
            flags |= FLAG_NAN_LOG_WEIGHT
            idx[b] = K
            continue
        m = row.max() if K else 0.0
        if not np.isfinite(m):
            flags |= FLAG_DEGENERATE_ROW
            idx[b] = K
            continue
        w = np.exp(row.astype(np.float64) - np.float64(m))
        c = np.cumsum(w)
        c = c / c[-1]
        pos = (u[b] + pos_k) / K
        idx[b] = np.searchsorted(c, pos, side="right")
    return idx, flags


def ancestor_index_reference_dtype(log_w, u):
    """The reference's own arithmetic (CDF in the input dtype through scipy logsumexp, np.exp,
    sequential np.cumsum, division by the row max; aesmc/inference.py:253-264, math.py:21-26,48-49)
    with the uniforms passed in instead of drawn.  Used to measure the float32 mismatch rate."""
    import scipy.special
    log_w = np.asarray(log_w)
    B, K = log_w.shape
    u = np.asarray(u, dtype=np.float64).reshape(B, 1)
    pos = (u + np.arange(0, K)) / K
    with np.errstate(all="ignore"):
        w = np.exp(log_w - scipy.special.logsumexp(log_w, axis=1, keepdims=True))
        c = np.cumsum(w, axis=1)
        c = c / np.max(c, axis=1, keepdims=True)
    idx = np.empty((B, K), dtype=np.int64)
    for b in range(B):
        idx[b] = np.digitize(pos[b], c[b])
    return idx


def ancestor_index_float32_cdf(log_w, u):
    """K2's opt-in mode for float32 rows (aesmc_set_float32_cdf(1)): `ancestor_index_reference_dtype` restated without
    SciPy — the row's logsumexp formed in float64 and rounded to float32 (SciPy returns the float32 of a pairwise
    float32 sum: the one step restated only to its result), np.exp on the float32 difference as the float32 of the
    float64 exponential, np.cumsum's sequential float32 sum, the float32 division by the last entry, float64
    positions (aesmc/inference.py:253-264, aesmc/math.py:21-26,48-49).  Returns (idx, flags) like `ancestor_index`."""
    log_w = np.asarray(log_w, dtype=np.float32)
    B, K = log_w.shape
    u = np.asarray(u, dtype=np.float64).reshape(B)
    idx = np.empty((B, K), dtype=np.int64)
    flags = 0
    pos_k = np.arange(0, K)
    for b in range(B):
        row = log_w[b]
        if np.isnan(row).any():
            flags |= FLAG_NAN_LOG_WEIGHT
            idx[b] = K
            continue
        m = np.float64(row.max()) if K else 0.0
        if not np.isfinite(m):
            flags |= FLAG_DEGENERATE_ROW
            idx[b] = K
            continue
        lse32 = np.float32(m + np.log(np.exp(row.astype(np.float64) - m).sum()))
        with np.errstate(under="ignore"):
            w32 = np.exp((row - lse32).astype(np.float64)).astype(np.float32)
        c32 = np.cumsum(w32, dtype=np.float32)           # left to right, every partial sum rounded to float32
        c32 = c32 / c32[-1]
        pos = (u[b] + pos_k) / K
        idx[b] = np.searchsorted(c32.astype(np.float64), pos, side="right")
    return idx, flags


def normal_logprob_sum(value, loc, scale):
    """K4: torch.distributions.Normal(loc, scale).log_prob(value) in PyTorch's operation order
    (normal.py log_prob: -((v - mu)**2) / (2 * var) - log(scale) - log(sqrt(2 pi))), summed over
    all dims past the first two as aesmc/state.py:151 does.  loc / scale broadcast to value."""
    value = np.asarray(value)
    dtype = value.dtype
    loc = np.broadcast_to(np.asarray(loc, dtype=dtype), value.shape)
    scale = np.broadcast_to(np.asarray(scale, dtype=dtype), value.shape)
    diff = value - loc
    var = scale * scale
    with np.errstate(all="ignore"):
        logp = (-(diff * diff)) / (dtype.type(2) * var) - np.log(scale) - dtype.type(0.9189385332046727)
    return logp.reshape(value.shape[0], value.shape[1], -1).sum(axis=2, dtype=dtype)


def normal_logprob_sum_backward(value, loc, scale, grad_out):
    """K4 backward: dense gradients (d/dvalue, d/dloc, d/dscale), each of value's shape."""
    value = np.asarray(value)
    dtype = value.dtype
    loc = np.broadcast_to(np.asarray(loc, dtype=dtype), value.shape)
    scale = np.broadcast_to(np.asarray(scale, dtype=dtype), value.shape)
    g = np.asarray(grad_out, dtype=dtype).reshape(value.shape[:2] + (1,) * (value.ndim - 2))
    diff = value - loc
    var = scale * scale
    gz = g * (diff / var)
    return -gz, gz, g * ((diff * diff) / (var * scale) - dtype.type(1) / scale)


def gather(src, idx):
    """K3: dst[b,k,...] = src[b, idx[b,k], ...] (torch.gather of aesmc/state.py:179).  Indices
    outside [0,K) are clamped and reported.  Returns (dst, flags)."""
    B, K = idx.shape
    flags = 0
    if ((idx < 0) | (idx >= K)).any():
        flags |= FLAG_INDEX_OUT_OF_RANGE
    safe = np.clip(idx, 0, max(K - 1, 0))
    rows = np.arange(B)[:, None]
    return np.array(src[rows, safe], copy=True), flags


def gather_backward(grad_out, idx):
    """K3 backward: grad_src[b,j,...] = sum_{k: idx[b,k]==j} grad_out[b,k,...].  Out-of-range
    indices contribute nothing and are reported."""
    B, K = idx.shape
    flags = 0
    bad = (idx < 0) | (idx >= K)
    if bad.any():
        flags |= FLAG_INDEX_OUT_OF_RANGE
    grad_src = np.zeros_like(grad_out)
    rows = np.broadcast_to(np.arange(B)[:, None], idx.shape)
    good = ~bad
    np.add.at(grad_src, (rows[good], idx[good]), grad_out[good])
    return grad_src, flags


def normal_rsample(eps, loc, scale):
    """K6: loc + eps * scale with the product rounded before the sum, in eps's dtype — the two
    eager ops of torch/distributions/normal.py rsample (reached from aesmc/state.py:98-104)."""
    eps = np.asarray(eps)
    product = (eps * np.broadcast_to(np.asarray(scale, dtype=eps.dtype), eps.shape)).astype(eps.dtype)
    return (np.broadcast_to(np.asarray(loc, dtype=eps.dtype), eps.shape) + product).astype(eps.dtype)


def particle_summary(log_w, value=None):
    """K7 contract in float64: (log_ess [B], mean [B,...], second moment [B,...]) under
    w = softmax(log_w, axis=1) — aesmc/statistics.py:47-60 (mean), :63-76 (second moment; the
    variance is second - mean**2), :79-91 (log ESS = 2 lse(lw) - lse(2 lw)).  Rows without
    normalisable weights (NaN, max = +-inf) give NaN."""
    lw = np.asarray(log_w, dtype=np.float64)
    B, K = lw.shape
    with np.errstate(invalid="ignore", over="ignore"):
        m = lw.max(axis=1, keepdims=True)
        e = np.exp(lw - m)
        s1, s2 = e.sum(axis=1), (e * e).sum(axis=1)
        broken = ~np.isfinite(m[:, 0])
        log_ess = np.where(broken, np.nan, 2 * np.log(s1) - np.log(s2))
        mean = second = None
        if value is not None:
            v = np.asarray(value, dtype=np.float64)
            w = (e / s1[:, None]).reshape((B, K) + (1,) * (v.ndim - 2))
            mean, second = (w * v).sum(axis=1), (w * v * v).sum(axis=1)
            mean[broken] = np.nan
            second[broken] = np.nan
    return log_ess, mean, second
