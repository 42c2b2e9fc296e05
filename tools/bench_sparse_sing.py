"""replaid.sing on a sparse single-cell matrix (10x-like: 33,538 genes, ~2,000 stored values per cell) through the two host
entries: plaidhip_sing_csc (CSC slots to the device, dense min-ranks from the ranks of the stored values) against
plaidhip_sing_dense on the densified matrix (what the reference's R code does before ranking).
    python3 tools/bench_sparse_sing.py [--genes 33538 --cells 10000 --sets 5000 --per-cell 2000]"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=33538)
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--sets", type=int, default=5000)
    ap.add_argument("--per-cell", type=int, default=2000)
    a = ap.parse_args()
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = a.genes, a.cells, a.sets
    rng = np.random.default_rng(0)
    Gp, Gi = sy.geneset_csc(g, m)
    cols = []
    indptr = np.zeros(n + 1, dtype=np.int64)
    for c in range(n):
        k = int(rng.integers(a.per_cell // 2, a.per_cell * 3 // 2))
        cols.append(np.sort(rng.choice(g, size=k, replace=False)))
        indptr[c + 1] = indptr[c] + k
    Xi = np.concatenate(cols).astype(np.int32)
    Xx = np.round(rng.gamma(2.0, 1.5, size=len(Xi)), 1) + 0.1
    Xs = sp.csc_matrix((Xx, Xi, indptr), shape=(g, n))
    ctx = plaid_amd.Context(0)
    S1 = ctx.sing_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        S1 = ctx.sing_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi)
        t.append(time.perf_counter() - t0)
    print(f"{g} genes x {n} cells ({Xs.nnz / n:.0f} stored per cell) x {m} sets")
    print(f"sing_csc   (CSC to the device)          : {min(t) * 1e3:8.1f} ms  -> {m * n / min(t):.3e} scores/s incl. PCIe")
    t0 = time.perf_counter()
    D = np.asfortranarray(Xs.toarray())
    t_dens = time.perf_counter() - t0
    S2 = ctx.sing_dense(D, Gp, Gi)
    t = []
    for _ in range(2):
        t0 = time.perf_counter()
        S2 = ctx.sing_dense(D, Gp, Gi)
        t.append(time.perf_counter() - t0)
    print(f"sing_dense (densified on the host first): {min(t) * 1e3:8.1f} ms  (+ {t_dens * 1e3:.0f} ms to densify {D.nbytes / 1e9:.1f} GB on the host)")
    print(f"max abs difference between the two: {np.abs(S1 - S2).max():.2e}")


if __name__ == "__main__":
    main()
