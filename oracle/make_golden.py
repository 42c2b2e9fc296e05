"""Generate the committed golden fixtures under tests/golden/ (TEST INFRASTRUCTURE).

Run in the BUILD container only (`python oracle/make_golden.py`): it reads the reference's
bundled data files from /root/reference/inst/extdata (GPL-3, (c) BigOmics Analytics SA --
`hallmarks.gmt` is MSigDB hallmark sets, `pbmc3k-50cells.rda` is made by dev/extdata.R:1-15)
and writes DATA ONLY: inputs as plain arrays and expected outputs computed by
oracle/plaid_oracle.py.  Nothing under /root/reference is needed at test time.

  pbmc3k50.npz       the 7728 x 50 dgCMatrix as CSC arrays + dimnames + celltype
  hallmarks.gmt      verbatim copy of the reference's data file (input data)
  pbmc3k50_expected.npz   oracle outputs on that pair: gmt2mat pattern, plaid raw/normalised,
                     replaid.sing, replaid.ssgsea(alpha 0 / 0.25), sparse_colranks, colranks
  synthetic_cases.npz     small seeded matrices covering ties, negatives, +-0, constant and
                     all-zero columns, empty / singleton sets, exact-zero scores, both
                     branches of ignore.zero, and a forced-chunk crossprod
"""
from __future__ import annotations

import os
import shutil
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import plaid_oracle as po  # noqa: E402
from rda_reader import dgcmatrix_to_csc, read_rda  # noqa: E402

REF = "/root/reference/inst/extdata"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def fixture_pair():
    d = read_rda(os.path.join(REF, "pbmc3k-50cells.rda"))
    p, i, x, dim, rn, cn = dgcmatrix_to_csc(d["X"])
    np.savez_compressed(os.path.join(OUT, "pbmc3k50.npz"), p=p, i=i, x=x, dim=np.asarray(dim),
                        rownames=np.asarray(rn), colnames=np.asarray(cn),
                        celltype=np.asarray(d["celltype"]))
    shutil.copyfile(os.path.join(REF, "hallmarks.gmt"), os.path.join(OUT, "hallmarks.gmt"))
    X = sp.csc_matrix((x, i, p), shape=dim)
    names, gsets = po.read_gmt(os.path.join(OUT, "hallmarks.gmt"))
    D, grn, gcn = po.gmt2mat(names, gsets)
    Xd = np.asarray(X.todense())
    exp = dict(
        G_p=D.indptr.astype(np.int32), G_i=D.indices.astype(np.int32), G_dim=np.asarray(D.shape),
        G_rownames=np.asarray(grn), G_colnames=np.asarray(gcn),
        plaid_raw=po.plaid(X, rn, D, grn, normalize=False),
        plaid_norm=po.plaid(X, rn, D, grn, normalize=True),
        plaid_sum_raw=po.plaid(X, rn, D, grn, stats="sum", normalize=False),
        plaid_dense_norm=po.plaid(Xd, rn, D, grn, normalize=True),
        sing=po.replaid_sing(X, rn, D, grn),
        ssgsea_a0=po.replaid_ssgsea(X, rn, D, grn, alpha=0.0),
        ssgsea_a025=po.replaid_ssgsea(X, rn, D, grn, alpha=0.25),
        ssgsea_dense_a0=po.replaid_ssgsea(Xd, rn, D, grn, alpha=0.0),
        ssgsea_dense_a025=po.replaid_ssgsea(Xd, rn, D, grn, alpha=0.25),
        sparse_colranks_avg=po.sparse_colranks(X).data,
        sparse_colranks_min=po.sparse_colranks(X, ties_method="min").data,
        colranks_avg=po.colranks(X, ties_method="average").astype(np.float32),   # half-integers: exact in f32
        colranks_min=po.colranks(X, ties_method="min").astype(np.float32),
    )
    np.savez_compressed(os.path.join(OUT, "pbmc3k50_expected.npz"), **exp)


def synthetic_cases():
    rng = np.random.Generator(np.random.PCG64(7))
    out = {}
    # --- rank cases: 300 x 12, heavy ties, negatives, +-0, constant column, all-zero column
    g, n = 300, 12
    X = np.round(rng.normal(0, 1.5, size=(g, n)), 1)
    X[rng.random((g, n)) < 0.3] = 0.0
    X[::7, 1] = -0.0
    X[:, 4] = 2.5          # constant
    X[:, 5] = 0.0          # all zero
    X[:, 6] = rng.normal(size=g)  # tie-free
    out["rank_X"] = X
    for tm in ("average", "min", "max"):
        out[f"rank_{tm}"] = po.colranks(X, ties_method=tm)
        out[f"rank_signed_{tm}"] = po.colranks(X, signed=True, ties_method=tm)
    Xs = sp.csc_matrix(X)
    out["rank_csc_p"], out["rank_csc_i"], out["rank_csc_x"] = Xs.indptr, Xs.indices, Xs.data
    for tm in ("average", "min", "max"):
        out[f"rank_csc_{tm}"] = po.sparse_colranks(Xs, ties_method=tm).data
        out[f"rank_csc_signed_{tm}"] = po.sparse_colranks(Xs, signed=True, ties_method=tm).data

    # --- crossprod / normalise cases: 200 genes, 40 samples, 23 sets incl. empty + singleton
    g, n, m = 200, 40, 23
    Xc = rng.normal(8, 2, size=(g, n))
    sets = []
    for j in range(m):
        k = int(rng.integers(2, 60))
        sets.append(np.sort(rng.choice(g, size=k, replace=False)))
    sets[3] = np.zeros(0, dtype=np.int64)       # empty set -> score exactly 0
    sets[9] = np.asarray([17])                  # singleton
    Gp = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int32)
    Gi = np.concatenate(sets).astype(np.int32)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [f"g{k}" for k in range(g)]
    out["cp_X"], out["cp_Gp"], out["cp_Gi"] = Xc, Gp, Gi
    out["cp_mean_raw"] = po.plaid(Xc, rn, G, rn, normalize=False)
    out["cp_sum_raw"] = po.plaid(Xc, rn, G, rn, stats="sum", normalize=False)
    out["cp_mean_norm"] = po.plaid(Xc, rn, G, rn, normalize=True)      # has exact zeros (empty set) -> ignore.zero
    keep = [j for j in range(m) if j != 3]
    out["cp_nozero_norm"] = po.plaid(Xc, rn, G[:, keep], rn, normalize=True)   # no zeros -> plain medians
    Xneg = Xc - 8.0
    out["cp_neg_norm"] = po.plaid(Xneg, rn, G, rn, normalize=True)     # negatives + zeros: min != 0
    # forced small chunk (R/plaid.R:110-119)
    Gs = sp.csc_matrix(G @ sp.diags(1.0 / (1e-8 + np.asarray(G.sum(axis=0)).ravel())))
    out["cp_chunk7"] = po.chunked_crossprod(Gs, Xc, chunk=7)
    # normalize_medians direct, explicit flags, a column that is entirely zero
    S = rng.normal(0, 1, size=(15, 9))
    S[rng.random(S.shape) < 0.2] = 0.0
    S[:, 2] = 0.0
    out["nm_S"] = S
    out["nm_auto"] = po.normalize_medians(S)[0]
    out["nm_true"] = po.normalize_medians(S, True)[0]
    out["nm_false"] = po.normalize_medians(S, False)[0]
    Sp = np.abs(S)
    out["nm_pos_S"] = Sp
    out["nm_pos_auto"] = po.normalize_medians(Sp)[0]                   # min == 0 -> ignore.zero TRUE
    np.savez_compressed(os.path.join(OUT, "synthetic_cases.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    fixture_pair()
    synthetic_cases()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
