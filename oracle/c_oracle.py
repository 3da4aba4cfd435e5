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


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
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
