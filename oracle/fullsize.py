"""Checker for FULL-SIZE launches (TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's parity legs).

A full-size score matrix (5e8 ... 6e9 scores) cannot be recomputed on the CPU, but any COLUMN of it can: every
column of `plaid()` depends on its own sample column plus three global scalars (R/plaid.R:107, :556-557, :572,
:251).  So the checker

  1. picks probe columns of the launch that was actually timed -- the first ones, the ones whose elements sit either
     side of element offset 2^31 of the column-major result (the limit `chunked_crossprod` exists for,
     R/plaid.R:103-104), and the last ones;
  2. recomputes those columns phase by phase with the plain-C oracle (ranks -> crossprod -> medians);
  3. verifies the global scalars on the host from the device's own per-column vectors (`mean(medx)` from `med[]`,
     `max(rX)` from `colmax[]`), so no scalar is taken on trust;
  4. compares: ranks bit-exact, medians and scores within the north-star tolerance (reported: the actual error).

Nothing here touches the product path; the arrays it receives were copied off the device by the caller.
"""
from __future__ import annotations

import numpy as np

from . import c_oracle

RTOL, ATOL = 1e-5, 1e-9        # BASELINE.json north_star: scores within 1e-5 relative (atol: scores centred on 0)


def probe_columns(n: int, m: int, width: int = 256):
    """sorted unique column indices: first `width`, `width` around element offset 2^31 of an m x n column-major
    matrix (when it has one), last `width`.  Returns (cols int64[k], crosses_2_31 bool)."""
    width = max(1, min(width, n))
    blocks = [np.arange(0, width), np.arange(n - width, n)]
    crosses = m > 0 and (m * n) > 2**31
    if crosses:
        c = 2**31 // m                                  # the column that holds element 2^31
        lo = max(0, min(n - width, c - width // 2))
        blocks.append(np.arange(lo, lo + width))
    cols = np.unique(np.concatenate(blocks)).astype(np.int64)
    return cols, bool(crosses)


def contiguous_runs(cols):
    """[(lo, hi)) runs of a sorted index array (device-side slicing by run keeps the copies small)"""
    runs, start, prev = [], int(cols[0]), int(cols[0])
    for c in cols[1:]:
        c = int(c)
        if c != prev + 1:
            runs.append((start, prev + 1))
            start = c
        prev = c
    runs.append((start, prev + 1))
    return runs


def _err(got, exp):
    with np.errstate(all="ignore"):
        d = np.abs(got - exp)
        return float(np.nanmax(d)) if d.size else 0.0, float(np.nanmax(d / np.maximum(np.abs(exp), 1e-9))) if d.size else 0.0


def check_normalised(raw_oracle, s_final_gpu, med_gpu_all, cols, red_gpu=None, flags_gpu=None, minmax_full=None):
    """raw_oracle: oracle's un-normalised scores of the probe columns (m x k).  s_final_gpu: the same columns of the
    launch's normalised result.  med_gpu_all: the device's med[] for ALL n columns.  red_gpu: device {sum, count} of
    the medians.  flags_gpu: device {has_neg, has_zero, has_nan}.  minmax_full: (min, any_zero) of the launch's FULL
    un-normalised result from an independent device reduction, when the caller has one.
    Returns a dict of errors; raises AssertionError beyond the tolerance."""
    out = {}
    # min(x) == 0 (R/plaid.R:556-557) is a property of the whole matrix: taken from the independent reduction when
    # there is one, else from the flags -- which must at least not contradict the probe columns
    if minmax_full is not None:
        mn, anyzero = minmax_full
        iz = bool(mn == 0.0)
        if flags_gpu is not None:
            assert bool(flags_gpu[0]) == bool(mn < 0.0), "has_neg flag disagrees with min(S) of the full launch"
            assert bool(flags_gpu[1]) == bool(anyzero), "has_zero flag disagrees with the full launch"
    else:
        assert flags_gpu is not None
        iz = (not bool(flags_gpu[0])) and bool(flags_gpu[1])
    if flags_gpu is not None:
        assert bool(flags_gpu[0]) or not (raw_oracle < 0).any(), "negative scores in the probe columns but has_neg unset"
        assert bool(flags_gpu[1]) or not (raw_oracle == 0).any(), "zero scores in the probe columns but has_zero unset"
    out["ignore_zero"] = iz
    _, med_o = c_oracle.normalize_medians_mt(np.asfortranarray(raw_oracle.copy()), iz, _threads())
    med_g = np.asarray(med_gpu_all, dtype=np.float64)
    a, r = _err(med_g[cols], med_o)
    out["med_max_abs_err"], out["med_max_rel_err"] = a, r
    np.testing.assert_allclose(med_g[cols], med_o, rtol=RTOL, atol=ATOL)
    mean_host = float(np.nanmean(med_g))                               # mean(medx, na.rm = TRUE), :572
    if red_gpu is not None:
        mean_dev = float(red_gpu[0]) / float(red_gpu[1])
        out["mean_med_rel_err"] = abs(mean_dev - mean_host) / max(abs(mean_host), 1e-300)
        assert out["mean_med_rel_err"] < 1e-12, (mean_dev, mean_host)
    expect = (raw_oracle - med_o[None, :]) + mean_host
    a, r = _err(s_final_gpu, expect)
    out["max_abs_err_vs_oracle"], out["max_rel_err_vs_oracle"] = a, r
    np.testing.assert_allclose(s_final_gpu, expect, rtol=RTOL, atol=ATOL)
    return out


def _threads():
    import os
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def ranks_from_powered(Rpow, power):
    """the half-integer ranks behind rank^power as the device stored them (error ~1e-15 relative, ranks <= 20,448:
    rounding to the nearest half is unambiguous)"""
    r = np.power(Rpow, 1.0 / power) if power != 1.0 else np.asarray(Rpow)
    return np.round(2.0 * r) / 2.0


def ssgsea_dense_raw(Xc, Gp, Gi, alpha, gmax):
    """oracle, R/plaid.R:246-253 on probe columns of a dense X with the GLOBAL max(rX) supplied (it couples all
    columns).  Returns (ranks, raw scores before normalize_medians)."""
    nt = _threads()
    R = c_oracle.colranks_dense_mt(Xc, "average", False, nt)
    W = R ** (1.0 + alpha)
    W /= gmax
    W -= 0.5
    return R, c_oracle.crossprod_dense(W, Gp, Gi, "mean", nt)


def ssgsea_csc_raw(Xp, Xi, Xx, g, Gp, Gi, alpha, gmax):
    """the same for a dgCMatrix (sparse_colranks of the stored values, R/plaid.R:631-650; the "- 0.5" of :251
    reaches every gene of a set, stored or not)"""
    nt = _threads()
    r = c_oracle.sparse_colranks_mt(Xp, Xx, "average", False, nt)
    w = r ** (1.0 + alpha)
    k = np.diff(Gp).astype(np.float64)
    S = c_oracle.crossprod_csc(Xp, Xi, w, g, Gp, Gi, "mean", nt)
    S /= gmax
    S -= (0.5 * k / (1e-8 + k))[:, None]
    return r, S


def sub_csc(runs):
    """one CSC (p, i, x) over the concatenated column runs of a larger CSC matrix; per run the caller passes
    (p[lo:hi+1], i[p[lo]:p[hi]], x[p[lo]:p[hi]]) -- the slices it copied off the device"""
    ps, idx, val, base = [np.zeros(1, dtype=np.int64)], [], [], 0
    for (p, i, x) in runs:
        p = np.asarray(p, dtype=np.int64)
        ps.append(p[1:] - p[0] + base)
        base += int(p[-1] - p[0])
        idx.append(np.asarray(i))
        val.append(np.asarray(x))
    return (np.concatenate(ps).astype(np.int32), np.concatenate(idx).astype(np.int32),
            np.concatenate(val).astype(np.float64))
