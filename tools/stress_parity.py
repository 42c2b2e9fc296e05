#!/usr/bin/python3
"""Randomised parity stress (GPU): many random shapes / densities / value patterns through the C ABI against the
oracle.  `run_case` is collected by pytest (tests/test_gpu_parity.py::test_stress_parity_random_case, 200 seeds);
standalone for longer runs through gpurun:  python tools/stress_parity.py --cases 1500"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import scipy.sparse as sp


def run_case(ctx, seed):
    """one random case through the host entry points; returns a list of failure strings (empty = parity)"""
    from oracle import plaid_oracle as po
    rng = np.random.default_rng(seed)
    out = []
    g = int(rng.choice([7, 64, 300, 2049, 8193, 10224, 10225, 16001, 20352, 20353, 20448, 20449, 26000, 36601, 51000]))
    n = int(rng.integers(1, 9))
    m = int(rng.choice([1, 3, 64, 65, 200, 1500, 3000, 5000, 6100, 21000])) if g >= 2049 else int(rng.integers(1, 80))
    kmax = int(min(g, rng.choice([3, 40, 400])))
    sizes = rng.integers(0, kmax + 1, size=m)
    sets = [np.sort(rng.choice(g, size=int(k), replace=False)) for k in sizes]
    Gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    Gi = (np.concatenate(sets) if Gp[-1] else np.zeros(0)).astype(np.int32)
    kind = rng.choice(["normal", "counts", "centred", "ties", "nan"])
    if kind == "normal":
        X = rng.normal(8, 2, size=(g, n))
    elif kind == "counts":
        X = np.where(rng.random((g, n)) < 0.08, np.round(rng.gamma(2, 1.5, size=(g, n)), 1), 0.0)
    elif kind == "centred":
        X = rng.normal(0, 1, size=(g, n))
    elif kind == "ties":
        X = rng.integers(-2, 3, size=(g, n)).astype(float)
    else:
        X = rng.normal(1, 1, size=(g, n))
        X[rng.random((g, n)) < 0.01] = 0.0
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    mode = str(rng.choice(["f64", "f64", "mixed"]))
    ctx.set_precision(mode)
    ctx.set_option("rank_kernel", str(rng.choice(["auto", "bucket", "network"])))
    rtol, atol = (1e-9, 1e-11) if mode == "f64" else (1e-5, 1e-6)
    checks = []
    tag = f"seed {seed} g={g} n={n} m={m} kind={kind} mode={mode}"
    try:
        norm = bool(rng.integers(0, 2))
        stat = str(rng.choice(["mean", "sum"]))
        checks.append(("plaid_dense", ctx.plaid_dense(X, Gp, Gi, stat, norm), po.plaid(X, rn, G, rn, stats=stat, normalize=norm)))
        Xs = sp.csc_matrix(X)
        for sm in ("scatter", "gather"):
            ctx.set_option("spmm_sparse_kernel", sm)
            checks.append(("plaid_csc_" + sm, ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, stat, norm),
                           po.plaid(Xs, rn, G, rn, stats=stat, normalize=norm)))
        ctx.set_option("spmm_sparse_kernel", "auto")
        if kind != "nan":
            checks.append(("sing", ctx.sing_dense(X, Gp, Gi), po.replaid_sing(X, rn, G, rn)))
            checks.append(("ssgsea", ctx.ssgsea_dense(X, Gp, Gi, 0.25), po.replaid_ssgsea(X, rn, G, rn, alpha=0.25)))
            tm = str(rng.choice(["average", "min", "max"]))
            R = ctx.colranks_dense(X, tm, False)
            if not np.array_equal(R, po.colranks(X, ties_method=tm)):
                out.append(f"{tag}: RANK MISMATCH ties={tm}")
            if Xs.nnz:
                Rs = ctx.colranks_csc(Xs.indptr, Xs.data, tm, False)
                if not np.array_equal(Rs, po.sparse_colranks(Xs, ties_method=tm).data):
                    out.append(f"{tag}: SPARSE RANK MISMATCH ties={tm}")
            # dense ranks of the sparse form (zeros ranked: from the ranks of the stored values, or densified when a
            # column stores too many), signed or not; and replaid.sing on the CSC slots
            sg = bool(rng.integers(0, 2))
            Rd = ctx.colranks_csc_dense(Xs.indptr, Xs.indices, Xs.data, g, tm, sg)
            if not np.array_equal(Rd, po.colranks(X, signed=sg, ties_method=tm)):
                out.append(f"{tag}: DENSE-FROM-SPARSE RANK MISMATCH ties={tm} signed={sg}")
            checks.append(("sing_csc", ctx.sing_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi), po.replaid_sing(X, rn, G, rn)))
            # ties.method passed through (first / last / dense: composed from min-rank passes, R/plaid.R:593,614-617)
            tm2 = str(rng.choice(["first", "last", "dense"]))
            R2 = ctx.colranks_dense(X, tm2, sg)
            if not np.array_equal(R2, po.colranks(X, signed=sg, ties_method=tm2)):
                out.append(f"{tag}: RANK MISMATCH ties={tm2} signed={sg}")
            # the thin callers: replaid.scse (R/plaid.R:155-190) and replaid.ucell (:276-282), dense or sparse input
            Xin = Xs if bool(rng.integers(0, 2)) else X
            rl = [None, True, False][int(rng.integers(0, 3))]
            smn = bool(rng.integers(0, 2))
            checks.append(("scse", ctx.scse(Xin, Gp, Gi, rl, smn), po.replaid_scse(Xin, rn, G, rn, remove_log2=rl, score_mean=smn)))
            if m > 0 and int(sizes.max()) > 0:
                rmax = float(rng.choice([50, 1500]))
                checks.append(("ucell", ctx.ucell(Xin, Gp, Gi, sizes.astype(np.float64), rmax), po.replaid_ucell(Xin, rn, G, rn, rmax=rmax)))
        # t(x) %*% y with per-entry weights (chunked_crossprod's general case), y dense and sparse
        W = sp.csc_matrix((rng.normal(size=len(Gi)), Gi, Gp), shape=(g, m))
        Y = np.nan_to_num(X, nan=0.5)
        checks.append(("weighted_dense", ctx.crossprod_weighted(W.indptr, W.indices, W.data, g, Y=Y), po.chunked_crossprod(W, Y)))
        Ys = sp.csc_matrix(Y)
        checks.append(("weighted_csc", ctx.crossprod_weighted(W.indptr, W.indices, W.data, g, Yp=Ys.indptr, Yi=Ys.indices, Yx=Ys.data),
                       po.chunked_crossprod(W, Ys)))
        S = rng.normal(0, 1, size=(m, n))
        S[rng.random(S.shape) < 0.1] = 0.0
        checks.append(("normalize_medians", ctx.normalize_medians(S.copy())[0], po.normalize_medians(S.copy())[0]))
    except Exception as exc:
        out.append(f"{tag}: EXCEPTION {type(exc).__name__}: {exc}")
        return out
    finally:
        ctx.set_option("rank_kernel", "auto")
    for name, got, exp in checks:
        got = np.asarray(got)
        exp = np.asarray(exp)
        scale = max(1.0, float(np.nanmax(np.abs(exp)))) if exp.size else 1.0   # mixed mode: error ~ 6e-8 x the magnitude of the summed values
        ok = got.shape == exp.shape and np.allclose(got, exp, rtol=rtol, atol=atol * (scale if mode == 'mixed' else 1.0), equal_nan=True)
        if not ok:
            err = np.nanmax(np.abs(got - exp)) if got.shape == exp.shape else float("nan")
            out.append(f"{tag}: MISMATCH {name} max abs err {err:.3e}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import plaid_amd
    ctx = plaid_amd.Context(0)
    bad = 0
    t0 = time.time()
    for case in range(a.cases):
        for line in run_case(ctx, a.seed * 100003 + case):
            print(line)
            bad += 1
    ctx.close()
    print(f"{a.cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
