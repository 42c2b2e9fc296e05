"""ctypes binding of oracle/liboracle.so (plain-C restatement, TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
_TIES = {"average": 0, "min": 1, "max": 2}


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def plaid_dense(X, Gp, Gi, stat="mean", normalize=True):
    X = np.asfortranarray(X, dtype=np.float64)
    Gp = np.ascontiguousarray(Gp, dtype=np.int32)
    Gi = np.ascontiguousarray(Gi, dtype=np.int32)
    g, n = X.shape
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    load().oracle_plaid_dense(_p(X), C.c_int32(g), C.c_int32(n), _p(Gp), _p(Gi), C.c_int32(m),
                              C.c_int(stat == "sum"), C.c_int(bool(normalize)), _p(S))
    return S


def normalize_medians(S, ignore_zero=None):
    S = np.array(S, dtype=np.float64, order="F", copy=True)
    m, n = S.shape
    med = np.empty(n)
    iz = -1 if ignore_zero is None else int(bool(ignore_zero))
    load().oracle_normalize_medians(_p(S), C.c_int32(m), C.c_int32(n), C.c_int(iz), _p(med))
    return S, med


def colranks_dense(X, ties="average", signed=False):
    X = np.asfortranarray(X, dtype=np.float64)
    g, n = X.shape
    R = np.empty((g, n), dtype=np.float64, order="F")
    load().oracle_colranks_dense(_p(X), C.c_int32(g), C.c_int32(n), C.c_int(_TIES[ties]), C.c_int(bool(signed)), _p(R))
    return R


def sparse_colranks(Xp, Xx, ties="average", signed=False):
    Xp = np.ascontiguousarray(Xp, dtype=np.int32)
    Xx = np.ascontiguousarray(Xx, dtype=np.float64)
    R = np.empty(len(Xx))
    load().oracle_sparse_colranks(_p(Xp), _p(Xx), C.c_int32(len(Xp) - 1), C.c_int(_TIES[ties]),
                                  C.c_int(bool(signed)), _p(R))
    return R


# ---- multi-threaded variants (bench.py "all cores" baseline) and the sparse crossprod -------------------
def crossprod_csc(Xp, Xi, Xx, g, Gp, Gi, stat="mean", threads=1):
    """t(G) %*% X for a dgCMatrix X in Gustavson order (work ~ stored values x sets per gene)"""
    Xp = np.ascontiguousarray(Xp, dtype=np.int32)
    Xi = np.ascontiguousarray(Xi, dtype=np.int32)
    Xx = np.ascontiguousarray(Xx, dtype=np.float64)
    Gp = np.ascontiguousarray(Gp, dtype=np.int32)
    Gi = np.ascontiguousarray(Gi, dtype=np.int32)
    n, m = len(Xp) - 1, len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    load().oracle_crossprod_csc_mt(_p(Xp), _p(Xi), _p(Xx), C.c_int32(g), C.c_int32(n), _p(Gp), _p(Gi), C.c_int32(m),
                                   C.c_int(stat == "sum"), _p(S), C.c_int(threads))
    return S


def crossprod_dense(X, Gp, Gi, stat="mean", threads=1):
    X = np.asfortranarray(X, dtype=np.float64)
    Gp = np.ascontiguousarray(Gp, dtype=np.int32)
    Gi = np.ascontiguousarray(Gi, dtype=np.int32)
    g, n = X.shape
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    load().oracle_crossprod_dense_mt(_p(X), C.c_int32(g), C.c_int32(n), _p(Gp), _p(Gi), C.c_int32(m),
                                     C.c_int(stat == "sum"), _p(S), C.c_int(threads))
    return S


def colranks_dense_mt(X, ties="average", signed=False, threads=1):
    X = np.asfortranarray(X, dtype=np.float64)
    g, n = X.shape
    R = np.empty((g, n), dtype=np.float64, order="F")
    load().oracle_colranks_dense_mt(_p(X), C.c_int32(g), C.c_int32(n), C.c_int(_TIES[ties]), C.c_int(bool(signed)),
                                    _p(R), C.c_int(threads))
    return R


def sparse_colranks_mt(Xp, Xx, ties="average", signed=False, threads=1):
    Xp = np.ascontiguousarray(Xp, dtype=np.int32)
    Xx = np.ascontiguousarray(Xx, dtype=np.float64)
    R = np.empty(len(Xx))
    load().oracle_sparse_colranks_mt(_p(Xp), _p(Xx), C.c_int32(len(Xp) - 1), C.c_int(_TIES[ties]),
                                     C.c_int(bool(signed)), _p(R), C.c_int(threads))
    return R


def normalize_medians_mt(S, ignore_zero=None, threads=1):
    """in place on a Fortran-ordered float64 array"""
    assert S.flags.f_contiguous and S.dtype == np.float64
    m, n = S.shape
    med = np.empty(n)
    iz = -1 if ignore_zero is None else int(bool(ignore_zero))
    load().oracle_normalize_medians_mt(_p(S), C.c_int32(m), C.c_int32(n), C.c_int(iz), _p(med), C.c_int(threads))
    return S, med
