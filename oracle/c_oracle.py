"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/smc_core.c (the plain-C
restatement of the index / byte work of the path).  `load()` builds the library with the Makefile
beside it when it is missing.  Only tests/ may import this module."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsmc_oracle.so")
_lib = None
_i64, _dp, _ip, _bp = ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p
_fp = ctypes.POINTER(ctypes.c_float)


def load():
    global _lib
    if _lib is None:
        source = os.path.join(HERE, "smc_core.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(source):      # (stale: rebuilt)
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        lib = ctypes.CDLL(LIB)
        lib.smc_oracle_ancestor_index.restype = ctypes.c_int
        lib.smc_oracle_ancestor_index.argtypes = [_dp, _dp, _ip, _i64, _i64]
        lib.smc_oracle_gather.restype = ctypes.c_int
        lib.smc_oracle_gather.argtypes = [_bp, _ip, _bp, _i64, _i64, _i64]
        lib.smc_oracle_gather_backward.restype = ctypes.c_int
        lib.smc_oracle_gather_backward.argtypes = [_dp, _ip, _dp, _i64, _i64, _i64]
        lib.smc_oracle_lineage.restype = None
        lib.smc_oracle_lineage.argtypes = [_ip, _ip, _i64, _i64, _i64]
        lib.smc_oracle_logweight_lse.restype = None
        lib.smc_oracle_logweight_lse.argtypes = [_dp, _dp, _dp, _dp, _dp, _i64, _i64]
        for suffix, real, rp in (("f32", ctypes.c_float, _fp), ("f64", ctypes.c_double, _dp)):
            fn = getattr(lib, "smc_oracle_particle_affine_" + suffix)
            fn.restype = None
            fn.argtypes = [rp, rp, _i64, _i64, _i64, rp, rp, _i64, _i64, _i64, rp, _i64, rp, rp, _i64, _i64, _i64]
            fn = getattr(lib, "smc_oracle_affine_rsample_" + suffix)
            fn.restype = None
            fn.argtypes = [rp, rp, _i64, _i64, rp, _i64, rp, real, rp, _i64, _i64, _i64, _i64]
            fn = getattr(lib, "smc_oracle_affine_logweight_" + suffix)
            fn.restype = None
            fn.argtypes = [rp, rp, rp, _i64, rp, rp, _i64, rp, rp, _i64, rp, rp, _i64, real, real, real, rp,
                           _i64, _i64, _i64, _i64]
            fn = getattr(lib, "smc_oracle_particle_mlp_" + suffix)
            fn.restype = None
            fn.argtypes = [rp, rp, rp, _i64, rp, rp, rp, _i64, _i64, _i64, _i64, _i64]
        _lib = lib
    return _lib


def _d(array):
    return np.ascontiguousarray(array, dtype=np.float64)


def _ptr(array, kind):
    return None if array is None else array.ctypes.data_as(kind)


def ancestor_index(log_w, u):
    """(idx int64 [B,K], flags) — log_w of any float dtype is widened to float64 (exact)."""
    lw, u = _d(log_w), _d(u).reshape(-1)
    B, K = lw.shape
    idx = np.empty((B, K), dtype=np.int64)
    flags = load().smc_oracle_ancestor_index(_ptr(lw, _dp), _ptr(u, _dp), _ptr(idx, _ip), B, K)
    return idx, flags


def gather(src, idx):
    src = np.ascontiguousarray(src)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B, K = idx.shape
    dst = np.empty_like(src)
    row_bytes = src.dtype.itemsize * int(np.prod(src.shape[2:], dtype=np.int64))
    flags = load().smc_oracle_gather(src.ctypes.data, _ptr(idx, _ip), dst.ctypes.data, B, K, row_bytes)
    return dst, flags


def gather_backward(grad_out, idx):
    """float64 result whatever grad_out's dtype (the caller rounds for comparison)."""
    g = _d(grad_out)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B, K = idx.shape
    D = int(np.prod(g.shape[2:], dtype=np.int64))
    out = np.empty_like(g)
    flags = load().smc_oracle_gather_backward(_ptr(g, _dp), _ptr(idx, _ip), _ptr(out, _dp), B, K, D)
    return out, flags


def lineage(indices):
    """indices: list of T-1 arrays [B,K] -> list of T lineage arrays [B,K] (last = identity)."""
    if len(indices) == 0:
        raise ValueError("lineage needs at least one index array")
    stack = np.ascontiguousarray(np.stack(indices), dtype=np.int64)
    T = stack.shape[0] + 1
    B, K = stack.shape[1:]
    out = np.empty((T, B, K), dtype=np.int64)
    load().smc_oracle_lineage(_ptr(stack, _ip), _ptr(out, _ip), T, B, K)
    return list(out)


def logweight_lse(a, b=None, c=None):
    a, b, c = _d(a), (None if b is None else _d(b)), (None if c is None else _d(c))
    B, K = a.shape
    lw, lse = np.empty_like(a), np.empty(B, dtype=np.float64)
    load().smc_oracle_logweight_lse(_ptr(a, _dp), _ptr(b, _dp), _ptr(c, _dp), _ptr(lw, _dp), _ptr(lse, _dp), B, K)
    return lw, lse


# ---- linear-Gaussian particle propagation (K8 / K9 / K10): computed in the arrays' own dtype ---------
def _real(array, dtype):
    return None if array is None else np.ascontiguousarray(array, dtype=dtype)


def _kind(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", _fp
    if dtype == np.float64:
        return "f64", _dp
    raise TypeError("float32 or float64 expected, got {}".format(dtype))


def _offset(off, B, dout, dtype):
    """offset as (array or None, batch stride): [dout] is shared by every row (stride 0), [B, dout] is per row."""
    if off is None:
        return None, 0
    off = _real(off, dtype)
    if off.ndim == 1:
        assert off.shape == (dout,)
        return off, 0
    assert off.shape == (B, dout)
    return off, dout


def particle_affine(x1, w1, x2=None, w2=None, offset=None, base=None):
    """K8: base + (offset + x1 @ w1.T + x2 @ w2.T) by the kernels' fma chain; x* [B,K,d*], w* [dout,d*]."""
    dtype = x1.dtype
    suffix, rp = _kind(dtype)
    x1, w1 = _real(x1, dtype), _real(w1, dtype)
    B, K, d1 = x1.shape
    dout = w1.shape[0]
    x2, w2, base = _real(x2, dtype), _real(w2, dtype), _real(base, dtype)
    d2 = 0 if x2 is None else x2.shape[2]
    off, off_sb = _offset(offset, B, dout, dtype)
    out = np.empty((B, K, dout), dtype=dtype)
    getattr(load(), "smc_oracle_particle_affine_" + suffix)(
        _ptr(x1, rp), _ptr(w1, rp), d1, 1, d1, _ptr(x2, rp), _ptr(w2, rp), d2, 1, d2, _ptr(off, rp), off_sb,
        _ptr(base, rp), _ptr(out, rp), B, K, dout)
    return out


def affine_rsample(source, weight, offset, eps, scale):
    """K9: (offset + source @ weight.T) + eps * scale."""
    dtype = source.dtype
    suffix, rp = _kind(dtype)
    source, weight, eps = _real(source, dtype), _real(weight, dtype), _real(eps, dtype)
    B, K, din = source.shape
    dout = weight.shape[0]
    off, off_sb = _offset(offset, B, dout, dtype)
    out = np.empty((B, K, dout), dtype=dtype)
    getattr(load(), "smc_oracle_affine_rsample_" + suffix)(
        _ptr(source, rp), _ptr(weight, rp), din, 1, _ptr(off, rp), off_sb, _ptr(eps, rp), float(scale),
        _ptr(out, rp), B, K, dout, din)
    return out


def affine_logweight(x_prev, x, y, transition, emission, proposal, scale_p, scale_g, scale_q):
    """K10: each of transition / emission / proposal is (weight, offset or None); y is [B, dy]."""
    dtype = x.dtype
    suffix, rp = _kind(dtype)
    x_prev, x, y = _real(x_prev, dtype), _real(x, dtype), _real(y, dtype)
    B, K, dx = x.shape
    dy = y.shape[1]
    args = []
    for (weight, offset), dout in ((transition, dx), (emission, dy), (proposal, dx)):
        weight = _real(weight, dtype)
        assert weight.shape == (dout, dx)
        off, off_sb = _offset(offset, B, dout, dtype)
        args += [weight, off, off_sb]
    lw = np.empty((B, K), dtype=dtype)
    getattr(load(), "smc_oracle_affine_logweight_" + suffix)(
        _ptr(x_prev, rp), _ptr(x, rp), _ptr(y, rp), dy,
        _ptr(args[0], rp), _ptr(args[1], rp), args[2], _ptr(args[3], rp), _ptr(args[4], rp), args[5],
        _ptr(args[6], rp), _ptr(args[7], rp), args[8], float(scale_p), float(scale_g), float(scale_q),
        _ptr(lw, rp), B, K, dx, dy)
    return lw


def particle_mlp(x, weight1, offset1, weight2, bias2=None):
    """K13: bias2 + tanh(offset1 + x @ weight1.T) @ weight2.T; offset1 [H] or [B, H]."""
    dtype = x.dtype
    suffix, rp = _kind(dtype)
    x, weight1, weight2, bias2 = _real(x, dtype), _real(weight1, dtype), _real(weight2, dtype), _real(bias2, dtype)
    B, K, din = x.shape
    hid, dout = weight1.shape[0], weight2.shape[0]
    off, off_sb = _offset(offset1, B, hid, dtype)
    out = np.empty((B, K, dout), dtype=dtype)
    getattr(load(), "smc_oracle_particle_mlp_" + suffix)(
        _ptr(x, rp), _ptr(weight1, rp), _ptr(off, rp), off_sb, _ptr(weight2, rp), _ptr(bias2, rp), _ptr(out, rp),
        B, K, din, hid, dout)
    return out
