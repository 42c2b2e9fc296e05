#!/usr/bin/python3
"""Randomised soak of the medians selected inside the sparse crossprod launch (tests/test_gpu_fused_medians.py's harness on
random shapes): gene-set collections of both kinds, 4,200 ... 9,000 cells of varying density, counts or rank weights, mean or
sum, random subsets of empty / 30x denser / NaN-holding / outlier / duplicated cells.  Every case asserts that the fused
medians equal the standalone kernels' on the same score matrix bit for bit (inside `_run`) and that the scores equal the
plain route's where the sums are order-independent.  GPU:  python tools/soak_fused_medians.py --cases 40"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dense", action="store_true",
                    help="the DENSE route (round 5: plaidhip_dev_spmm_dense_fused_f64, pair kernel): random gene counts incl. odd ones "
                         "and one / two / three gene slices, leading dimensions, value kinds, odd columns")
    a = ap.parse_args()
    if a.dense:
        return main_dense(a)
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    from test_gpu_fused_medians import _run
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    ctx.set_option("spmm_sparse_kernel", "scatter")
    ctx.set_option("fused_medians", "on")
    bad = 0
    t0 = time.time()
    for case in range(a.cases):
        rng = np.random.default_rng(a.seed * 100003 + case)
        real = bool(rng.integers(0, 2))
        g = int(rng.choice([12010, 17713, 20000]))
        m = int(rng.choice([6200, 9000, 17409, 24000, 34817, 50000, 61459]))
        n = int(rng.integers(4200, 9001))
        dens = float(rng.choice([0.02, 0.05, 0.07, 0.1]))
        ranks = bool(rng.integers(0, 2))
        stat = "mean" if ranks else str(rng.choice(["mean", "sum"]))
        Gp, Gi = (sy.geneset_csc_real(g, m, seed=int(rng.integers(1, 1 << 30))) if real else
                  sy.geneset_csc(g, m, seed=int(rng.integers(1, 1 << 30)), kmax=int(rng.choice([60, 500]))))
        gs = ctx.geneset(g, Gp, Gi)
        j0 = int(rng.integers(0, 100000))
        Xp, Xi, Xx = sy.sparse_columns(g, j0, j0 + n, density=dens)
        Xp = Xp.astype(np.int64)
        # perturb random cells: rebuild the slots column by column for the few that change
        kinds = {}
        for c in rng.choice(n, size=int(rng.integers(0, 40)), replace=False):
            kinds[int(c)] = str(rng.choice(["empty", "dense", "nan", "outlier", "dup"]))
        if kinds:
            cols_i, cols_x = [], []
            for c in range(n):
                i_, x_ = Xi[Xp[c]:Xp[c + 1]], Xx[Xp[c]:Xp[c + 1]]
                k = kinds.get(c)
                if k == "empty":
                    i_, x_ = i_[:0], x_[:0]
                elif k == "dense":
                    i_ = np.sort(rng.choice(g, size=min(g, 30 * max(len(i_), 1)), replace=False)).astype(np.int32)
                    x_ = np.log1p(rng.geometric(0.12, size=len(i_)).astype(np.float64))
                elif k == "nan" and len(x_) and not ranks:
                    x_ = x_.copy(); x_[int(rng.integers(0, len(x_)))] = np.nan
                elif k == "outlier" and len(x_):
                    x_ = x_.copy(); x_[:3] *= 1000.0
                elif k == "dup" and c > 0:
                    i_, x_ = cols_i[c - 1], cols_x[c - 1]
                cols_i.append(i_); cols_x.append(x_)
            Xp = np.concatenate([[0], np.cumsum([len(v) for v in cols_i])]).astype(np.int64)
            Xi = np.concatenate(cols_i).astype(np.int32)
            Xx = np.concatenate(cols_x)
        tag = f"case {case}: g={g} m={m} n={n} dens={dens} real={real} ranks={ranks} stat={stat} odd cells={len(kinds)}"
        try:
            note = ""
            if ranks:
                S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, 1.0, -0.5, ranks=True)
                ok = np.array_equal(f1, f2)
                if bool(torch.equal(S1, S2)):
                    ok = ok and np.array_equal(m1, m2, equal_nan=True)
                else:
                    # a collection with a set of thousands of genes (kbits = 15) leaves rank weights too few bits for the
                    # 2^-40 bound: both routes take fp64 atomics, whose sums depend on the arrival order in the last bits
                    note = "  (fp64 atomics: last bits)"
                    ok = ok and bool(torch.allclose(S1, S2, rtol=1e-12, atol=1e-14)) and np.allclose(m1, m2, rtol=1e-11, atol=1e-13)
            else:
                iz = [None, True, False][int(rng.integers(0, 3))]
                note = f"  ignore.zero={iz}"
                S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, ignore_zero=iz, stat=stat)
                ok = np.array_equal(f1, f2)
                fixed = not np.isnan(Xx).any() and not any(k == "outlier" for k in kinds.values())
                if fixed:   # (order-independent fixed-point sums: the two routes agree to the bit)
                    ok = ok and bool(torch.equal(torch.nan_to_num(S1, nan=-7.0), torch.nan_to_num(S2, nan=-7.0))) \
                        and np.array_equal(m1, m2, equal_nan=True)
            print(("ok   " if ok else "FAIL ") + tag + f"  resolved {status.mean() if len(status) else float('nan'):.4f}" + note, flush=True)
            bad += 0 if ok else 1
        except AssertionError as exc:
            print("FAIL " + tag + f"  {str(exc)[:200]}", flush=True)
            bad += 1
        gs.close()
        del S1, S2
        torch.cuda.empty_cache()
    print(f"{a.cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    ctx.close()
    return 1 if bad else 0


def main_dense(a):
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    from test_gpu_fused_medians_dense import _run
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    ctx.set_option("fused_medians", "on")
    bad = 0
    t0 = time.time()
    for case in range(a.cases):
        rng = np.random.default_rng(a.seed * 100019 + case)
        real = bool(rng.integers(0, 2))
        g = int(rng.choice([3001, 8000, 10224, 10225, 12010, 15001, 17713, 20000, 20448, 20449, 25000, 33538]))
        m = int(rng.choice([6200, 9000, 17409, 24000, 34817, 50000]))
        n = int(rng.integers(1100, 2400))
        ld = g + int(rng.choice([0, 0, 1, 2, 7]))
        kind = str(rng.choice(["normal", "rounded", "counts", "ssgsea", "centred"]))
        stat = "mean" if kind == "ssgsea" else str(rng.choice(["mean", "sum"]))
        Gp, Gi = (sy.geneset_csc_real(g, m, seed=int(rng.integers(1, 1 << 30))) if real else
                  sy.geneset_csc(g, m, seed=int(rng.integers(1, 1 << 30)), kmax=int(rng.choice([60, 500]))))
        gs = ctx.geneset(g, Gp, Gi)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(rng.integers(1, 1 << 30)))
        with torch.cuda.stream(stream):
            X = torch.zeros((n, ld), dtype=torch.float64, device=dev)
            Z = torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen)
            if kind == "normal":
                X[:, :g] = Z * 2.0 + 8.0
            elif kind == "rounded":
                X[:, :g] = torch.round((Z * 2.0 + 8.0) * 10.0) / 10.0
            elif kind == "counts":
                X[:, :g] = torch.floor(torch.rand((n, g), dtype=torch.float64, device=dev, generator=gen) * 4.0) * \
                    (torch.rand((n, g), dtype=torch.float64, device=dev, generator=gen) < 0.08)
            elif kind == "centred":
                X[:, :g] = Z
            alpha, beta, div = 1.0, 0.0, None
            if kind == "ssgsea":
                X[:, :g] = Z * 2.0 + 8.0
                R = torch.zeros_like(X)
                colmax = torch.zeros(n, dtype=torch.float64, device=dev)
                div = torch.zeros(1, dtype=torch.float64, device=dev)
                if g <= ctx.limit("sparse_rank_column"):
                    ctx.dev_colranks_dense(X.data_ptr(), ld, g, n, R.data_ptr(), ld, "average", False, 1.25, colmax.data_ptr())
                    ctx.dev_max(colmax.data_ptr(), n, div.data_ptr())
                    X, beta = R, -0.5
                else:
                    div = None
            # odd columns: a NaN, an Inf, an all-equal column, a copy of the neighbour
            odd = {}
            for c in rng.choice(n, size=int(rng.integers(0, 12)), replace=False):
                odd[int(c)] = str(rng.choice(["nan", "inf", "const", "dup"]))
            for c, k in odd.items():
                if k == "nan":
                    X[c, int(rng.integers(0, g))] = float("nan")
                elif k == "inf" and kind != "ssgsea":
                    X[c, int(rng.integers(0, g))] = float("inf")
                elif k == "const":
                    X[c, :g] = 3.0
                elif k == "dup" and c > 0:
                    X[c] = X[c - 1]
        iz = [None, True, False][int(rng.integers(0, 3))]
        tag = f"case {case}: g={g} ld={ld} m={m} n={n} kind={kind} real={real} stat={stat} ignore.zero={iz} odd columns={len(odd)}"
        try:
            S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, X, ld, n, alpha, beta, div, iz, stat)
            same_nan = bool(torch.equal(torch.isnan(S1), torch.isnan(S2)))
            ok = token > 0 and same_nan and np.array_equal(f1, f2) and np.array_equal(m1, m2, equal_nan=True) and \
                bool(torch.equal(torch.nan_to_num(S1, nan=-7.0, posinf=1e300, neginf=-1e300), torch.nan_to_num(S2, nan=-7.0, posinf=1e300, neginf=-1e300)))
            print(("ok   " if ok else "FAIL ") + tag + f"  resolved {status.mean() if len(status) else float('nan'):.4f}", flush=True)
            bad += 0 if ok else 1
            del S1, S2
        except AssertionError as exc:
            print("FAIL " + tag + f"  {str(exc)[:200]}", flush=True)
            bad += 1
        gs.close()
        del X
        torch.cuda.empty_cache()
    print(f"{a.cases} dense cases, {bad} failures, {time.time() - t0:.1f} s")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
