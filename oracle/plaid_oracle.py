"""CPU oracle: a numpy/scipy restatement of bigomics/plaid's scoring hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker.  `plaid_amd/` never imports it.

What it restates.  The reference (R package `plaid`, /root/reference @ 2025-06-14)
is pure R; its arithmetic is delegated to un-vendored, un-pinned CRAN packages
(DESCRIPTION:26-31: Matrix, matrixStats, sparseMatrixStats, Rfast).  R is not
installed in the build container, so the reference can neither be run nor
compiled here.  Every function below cites the reference lines it follows;
third-party calls are restated from their documented semantics:

  Matrix::crossprod(x, y)            -> t(x) %*% y in float64
  matrixStats::colRanks(ties.method) -> ascending ranks, NaN-free input,
  sparseMatrixStats::colRanks           "average"/"min"/"max" == scipy.stats.rankdata
  base::rank(ties.method)            -> same
  matrixStats::colMedians(na.rm=T)   -> numpy.nanmedian (even count: mean of the
                                        two middle order statistics)
  Rfast::ttests(ina=)                -> Welch two-sample t-test per row

Pinning status.  The reference's own test-suite pins
nothing (tests/testthat/test-plaid.R:1-3 is `expect_equal(2*2, 4)`).  The only
numeric pins in the tree are the vignette outputs doc/plaid-vignette.html:798,
809,857-869; `tests/test_oracle_kat.py` checks this oracle against them
(`dim(matG)`, `dim(gsetX)`, `p.one`, `p.lm`, `p.meta` for six hallmark sets),
which pins plaid() + normalize_medians() (incl. the ignore-zero rule) and the
G^T.v crossprod.  The RANK path (colranks / sparse_colranks / replaid.*) has no
pin anywhere in the reference: for it, PARITY IS UNPINNED -- it is anchored only
on the documented semantics of rank()/colRanks() and cross-checked against
scipy.stats.rankdata and an independent O(n^2) counting definition.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.stats as st

INT_MAX = 2147483647  # .Machine$integer.max


# ---------------------------------------------------------------------------
# GMT I/O  (R/gmt-utils.R)
# ---------------------------------------------------------------------------
def _unique_keep_order(seq):
    seen = set()
    out = []
    for s in seq:
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def read_gmt(path, add_source=False, nrows=-1):
    """R/gmt-utils.R:99-125.  One line per set; '#' starts a comment line;
    tab-split; field 1 = name, field 2 = source, rest = genes (re-split on
    space/tab, line 116); drop "", "NA" and duplicates (`setdiff`, line 117).
    Returns (names, list of gene lists) -- a list, because names may repeat."""
    names, gsets = [], []
    with open(path, "r", encoding="utf-8") as fh:
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if "#" in line:  # read.csv(comment.char="#"), line 106
                line = line.split("#", 1)[0]
            if line.strip() == "":
                continue
            f = line.split("\t")
            name = f[0]
            source = f[1] if len(f) > 1 else "NA"
            genes = " ".join(f[2:]) if len(f) >= 3 else ""
            toks = [t for t in genes.replace("\t", " ").split(" ")]
            toks = _unique_keep_order(t for t in toks if t not in ("", "NA"))
            if add_source:
                name = f"{name} ({source})"
            names.append(name)
            gsets.append(toks)
            if 0 < nrows <= len(names):
                break
    return names, gsets


def gmt2mat(names, gsets, max_genes=-1, ntop=-1, bg=None):
    """R/gmt-utils.R:19-66.  Returns (D csc float64 0/1, rownames, colnames).

    sets ordered by decreasing size, stable (line 25); duplicated names dropped,
    first kept (26); optional head(ntop) (27); background = genes by decreasing
    frequency (31; ties in `table()` order = sorted names -- locale collation in
    R, code-point order here; row order never affects scores because plaid()
    matches rows by name); head(max.genes) (35); membership restricted to the
    background (36); rows finally re-ordered by decreasing row sum, stable (62).
    """
    order = sorted(range(len(gsets)), key=lambda k: -len(gsets[k]))
    names = [names[k] for k in order]
    gsets = [gsets[k] for k in order]
    keep, seen = [], set()
    for k, nm in enumerate(names):
        if nm not in seen:
            seen.add(nm)
            keep.append(k)
    names = [names[k] for k in keep]
    gsets = [gsets[k] for k in keep]
    if ntop > 0:
        gsets = [g[:ntop] for g in gsets]
    if bg is None:
        cnt = {}
        for g in gsets:
            for x in g:
                cnt[x] = cnt.get(x, 0) + 1
        bg = sorted(sorted(cnt), key=lambda x: -cnt[x])
    if max_genes < 0:
        max_genes = len(bg)
    gg = list(bg[:max_genes])
    pos = {g: k for k, g in enumerate(gg)}
    rows, cols = [], []
    for j, g in enumerate(gsets):
        for x in _unique_keep_order(g):
            if x in pos:
                rows.append(pos[x])
                cols.append(j)
    D = sp.csc_matrix((np.ones(len(rows)), (rows, cols)),
                      shape=(len(gg), len(names)), dtype=np.float64)
    D.sum_duplicates()
    D.data[:] = 1.0
    rs = np.asarray((D != 0).sum(axis=1)).ravel()
    ro = np.argsort(-rs, kind="stable")
    D = D[ro, :].tocsc()
    D.sort_indices()
    return D, [gg[k] for k in ro], names


# ---------------------------------------------------------------------------
# plaid()  (R/plaid.R:60-123, 554-575)
# ---------------------------------------------------------------------------
def _r_round(x):
    """R's round(): IEC 60559 half-to-even (R >= 4.0)."""
    return int(np.round(x))


def chunked_crossprod(x, y, chunk=None, _log=None):
    """R/plaid.R:100-123: t(x) %*% y, optionally in column chunks of y.
    x: genes x sets (sparse or dense), y: genes x samples (sparse or dense).
    Returns a dense float64 ndarray (the reference densifies right after, :81)."""
    ncx = x.shape[1]
    ncy = y.shape[1]
    if chunk is None or chunk < 0:
        chunk = _r_round(0.8 * INT_MAX / ncx)            # :103-104
    xt = x.T.tocsr() if sp.issparse(x) else np.asarray(x).T

    def cp(yy):
        r = xt @ yy
        return np.asarray(r.todense()) if sp.issparse(r) else np.asarray(r)

    if ncy < chunk:                                       # :107
        return cp(y)
    if _log is not None:
        _log.append(chunk)                                # message(), :109
    k = math.ceil(ncy / chunk)                            # :110
    out = np.full((ncx, ncy), np.nan)                     # :111
    for i in range(1, k + 1):                             # :115-119
        lo = (i - 1) * chunk
        hi = min(ncy, i * chunk)
        out[:, lo:hi] = cp(y[:, lo:hi])
    return out


def normalize_medians(x, ignore_zero=None):
    """R/plaid.R:554-575.  Returns (normalised matrix, medians)."""
    x = np.asarray(x, dtype=np.float64)
    if ignore_zero is None:
        ignore_zero = bool(np.nanmin(x) == 0)             # :556-557
    if ignore_zero:
        zx = x.copy()
        zx[x == 0] = np.nan                               # :562-563
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                medx = np.nanmedian(zx, axis=0)           # :565
        medx[np.isnan(medx)] = 0.0                        # :566
    else:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            medx = np.nanmedian(x, axis=0)                # :569
    nx = (x - medx[None, :]) + np.nanmean(medx)           # :572
    return nx, medx


def align(rownames_x, rownames_g):
    """intersect(rownames(X), rownames(matG)) -- order of X, unique (R/plaid.R:65).
    Returns (gg, index into X rows, index into G rows)."""
    pos_g = {}
    for k, nm in enumerate(rownames_g):
        pos_g.setdefault(nm, k)
    gg, ix, ig, seen = [], [], [], set()
    for k, nm in enumerate(rownames_x):
        if nm in pos_g and nm not in seen:
            seen.add(nm)
            gg.append(nm)
            ix.append(k)
            ig.append(pos_g[nm])
    return gg, np.asarray(ix, dtype=np.int64), np.asarray(ig, dtype=np.int64)


def plaid(X, rownames_x, matG, rownames_g, stats="mean", chunk=None, normalize=True):
    """R/plaid.R:60-87.  X genes x samples (ndarray or scipy sparse), matG genes x
    sets (scipy sparse or ndarray).  Returns dense sets x samples, or None when no
    feature overlaps (:66-69)."""
    if X.ndim == 1:
        X = X.reshape(-1, 1)                              # :63
    gg, ix, ig = align(rownames_x, rownames_g)            # :65
    if len(gg) == 0:
        return None                                       # :66-69
    Xa = X[ix, :]                                         # :71
    Ga = sp.csc_matrix(matG)[ig, :]                       # :72
    G = sp.csc_matrix((Ga != 0).astype(np.float64))       # :73
    if stats == "mean":
        sumG = 1e-8 + np.asarray(G.sum(axis=0)).ravel()   # :75
        G = G @ sp.diags(1.0 / sumG)                      # :76 colScale
    gsetX = chunked_crossprod(sp.csc_matrix(G), Xa, chunk=None)  # :80 (chunk arg ignored)
    gsetX = np.asarray(gsetX, dtype=np.float64)           # :81
    if normalize:
        gsetX, _ = normalize_medians(gsetX)               # :83
    return gsetX


# ---------------------------------------------------------------------------
# ranks  (R/plaid.R:589-650)
# ---------------------------------------------------------------------------
def _rank_vec(x, ties_method):
    """base::rank / matrixStats::colRanks on one NaN-free vector.  ties.method is passed through by the reference
    (R/plaid.R:614-617, 639-642): "first" breaks ties by position (scipy "ordinal"), "last" by reverse position
    (rank(c(1, 1, 1), ties.method = "last") is 3 2 1), "dense" leaves no gaps (matrixStats only)."""
    if len(x) == 0:
        return np.zeros(0)
    if ties_method == "first":
        return st.rankdata(x, method="ordinal").astype(np.float64)
    if ties_method == "last":
        return st.rankdata(np.asarray(x)[::-1], method="ordinal")[::-1].astype(np.float64)
    if ties_method == "random":
        raise ValueError("ties.method = 'random' has no deterministic oracle")
    return st.rankdata(x, method=ties_method).astype(np.float64)


def rank_by_counting(x, ties_method="average"):
    """Independent O(n^2) definition used to cross-check rankdata in tests:
    rank_min = 1 + #{x_j < x_i}; rank_max = #{x_j <= x_i}; average = mean."""
    x = np.asarray(x, dtype=np.float64)
    lt = (x[None, :] < x[:, None]).sum(axis=1)
    le = (x[None, :] <= x[:, None]).sum(axis=1)
    if ties_method == "min":
        return (lt + 1).astype(np.float64)
    if ties_method == "max":
        return le.astype(np.float64)
    return (lt + 1 + le) / 2.0


def sparse_colranks(X, signed=False, ties_method="average"):
    """R/plaid.R:631-650: rank the stored non-zeros of each CSC column among
    themselves; pattern unchanged, @x replaced (:645-646)."""
    if ties_method == "dense":
        raise ValueError("'arg' should be one of average, first, last, random, max, min")   # base::rank has no "dense"
    X = sp.csc_matrix(X).copy()
    X.sort_indices()
    out = X.data.astype(np.float64).copy()
    p = X.indptr
    for j in range(X.shape[1]):
        v = X.data[p[j]:p[j + 1]]
        if signed:                                        # :637-640
            out[p[j]:p[j + 1]] = np.sign(v) * _rank_vec(np.abs(v), ties_method)
        else:                                             # :642
            out[p[j]:p[j + 1]] = _rank_vec(v, ties_method)
    return sp.csc_matrix((out, X.indices.copy(), X.indptr.copy()), shape=X.shape)


def colranks(X, sparse=None, signed=False, keep_zero=False, ties_method="average"):
    """R/plaid.R:589-623.  Dense result (ndarray genes x samples) except for the
    sparse + keep.zero branch, which returns scipy csc (:600-601)."""
    if sparse is None:
        sparse = sp.issparse(X)                           # :595-596
    if sparse and keep_zero:
        return sparse_colranks(X, signed=signed, ties_method=ties_method)
    if sparse and ties_method not in ("max", "average", "min"):
        raise ValueError("'arg' should be one of max, average, min")   # sparseMatrixStats::colRanks (:605-608)
    D = np.asarray(X.todense()) if sp.issparse(X) else np.asarray(X, dtype=np.float64)
    out = np.empty(D.shape, dtype=np.float64)
    for j in range(D.shape[1]):
        if signed:                                        # :603-606 / :612-615
            out[:, j] = np.sign(D[:, j]) * _rank_vec(np.abs(D[:, j]), ties_method)
        else:                                             # :608 / :617
            out[:, j] = _rank_vec(D[:, j], ties_method)
    return out


# ---------------------------------------------------------------------------
# replaid.*  (R/plaid.R:155-309)
# ---------------------------------------------------------------------------
def _dense(M):
    return np.asarray(M.todense()) if sp.issparse(M) else np.asarray(M, dtype=np.float64)


def replaid_sing(X, rownames_x, matG, rownames_g):
    """R/plaid.R:213-219."""
    rX = colranks(X, ties_method="min")                   # :215
    rX = rX / X.shape[0] - 0.5                            # :216
    return plaid(rX, rownames_x, matG, rownames_g, normalize=False)  # :217


def replaid_ssgsea(X, rownames_x, matG, rownames_g, alpha=0.0):
    """R/plaid.R:244-255.  For sparse X the rank step is sparse_colranks (zeros stay
    0) and the `- 0.5` densifies (:251)."""
    rX = colranks(X, keep_zero=True, ties_method="average")   # :245
    rX = _dense(rX)
    if alpha != 0:
        rX = rX ** (1 + alpha)                            # :249
    rX = rX / np.max(rX) - 0.5                            # :251 (global max)
    return plaid(rX, rownames_x, matG, rownames_g, stats="mean", normalize=True)  # :253


def replaid_ucell(X, rownames_x, matG, rownames_g, rmax=1500):
    """R/plaid.R:276-282."""
    rX = _dense(colranks(X, ties_method="average"))       # :277
    rX = np.minimum(np.max(rX) - rX, rmax + 1)            # :278
    S = plaid(rX, rownames_x, matG, rownames_g)           # :279
    k = np.asarray((sp.csc_matrix(matG) != 0).sum(axis=0)).ravel()
    return 1 - S / rmax + ((k + 1) / (2 * rmax))[:, None]  # :280


def replaid_aucell(X, rownames_x, matG, rownames_g, auc_max_rank=None):
    """R/plaid.R:304-309."""
    if auc_max_rank is None:
        auc_max_rank = math.ceil(0.05 * X.shape[0])
    rX = _dense(colranks(X, ties_method="average"))       # :305
    ww = 1.08 * np.maximum((rX - (np.max(rX) - auc_max_rank)) / auc_max_rank, 0)  # :306
    return plaid(ww, rownames_x, matG, rownames_g, stats="mean")  # :307


def replaid_scse(X, rownames_x, matG, rownames_g, remove_log2=None, score_mean=False):
    """R/plaid.R:155-190."""
    Xd = X.copy()
    if remove_log2 is None:                               # :160-161, min/max over ALL entries
        if sp.issparse(Xd):
            has_implicit = Xd.nnz < Xd.shape[0] * Xd.shape[1]
            mn = min(Xd.data.min(), 0.0) if (has_implicit and Xd.nnz) else (Xd.data.min() if Xd.nnz else 0.0)
            mx = max(Xd.data.max(), 0.0) if (has_implicit and Xd.nnz) else (Xd.data.max() if Xd.nnz else 0.0)
        else:
            mn, mx = np.nanmin(Xd), np.nanmax(Xd)
        remove_log2 = bool(mn == 0 and mx < 20)
    if remove_log2:
        if sp.issparse(Xd):
            Xd.data = 2.0 ** Xd.data                      # :166
        else:
            Xd = np.where(Xd > 0, 2.0 ** Xd, Xd)          # :168-169
    absX = abs(Xd)
    if score_mean:
        sX = plaid(Xd, rownames_x, matG, rownames_g, stats="mean", normalize=False)
        sumx = np.asarray(absX.mean(axis=0)).ravel() + 1e-8   # :176
        return sX / sumx[None, :]
    sX = plaid(Xd, rownames_x, matG, rownames_g, stats="sum", normalize=False)
    sumx = np.asarray(absX.sum(axis=0)).ravel() + 1e-8    # :181
    return sX / sumx[None, :] * 100                       # :182


def replaid_gsva(X, rownames_x, matG, rownames_g, tau=0, rowtf="z"):
    """R/plaid.R:338-363."""
    Xd = _dense(X)
    if rowtf == "z":
        with np.errstate(all="ignore"):
            zX = (Xd - Xd.mean(axis=1, keepdims=True)) / (1e-8 + Xd.std(axis=1, ddof=1, keepdims=True))   # :343
    elif rowtf == "ecdf":
        zX = np.stack([(np.sum(r[None, :] <= r[:, None], axis=1)) / len(r) for r in Xd])                   # :346
    else:
        raise ValueError("Error: unknown row transform" + str(rowtf))
    rX = colranks(zX, signed=True, ties_method="average")       # :352
    rX = rX / np.max(np.abs(rX))                                  # :353
    if tau > 0:
        rX = np.sign(rX) * np.abs(rX) ** (1 + tau)                # :357
    return plaid(rX, rownames_x, matG, rownames_g)                # :360


# ---------------------------------------------------------------------------
# plaid.test pieces -- only what the vignette known answers need
# (R/plaid.R:392-537)
# ---------------------------------------------------------------------------
def matrix_onesample_ttest(F, G):
    """R/plaid.R:476-486, F a vector (one column)."""
    Gb = sp.csc_matrix((sp.csc_matrix(G) != 0).astype(np.float64))
    sumG = np.asarray(Gb.sum(axis=0)).ravel()
    sum_sq = Gb.T @ (F ** 2)
    meanx = (Gb.T @ F) / (1e-8 + sumG)
    with np.errstate(all="ignore"):
        sdx = np.sqrt((sum_sq - meanx ** 2 * sumG) / (sumG - 1))
        t = meanx / (1e-8 + sdx) * np.sqrt(sumG)
    p = 2 * st.t.sf(np.abs(t), df=np.maximum(sumG - 1, 1))
    return meanx, t, p


def matrix_twosample_ttest(F, G):
    """R/plaid.R:488-520, F a vector (one column): genes of the set against all other genes."""
    Gb = sp.csc_matrix((sp.csc_matrix(G) != 0).astype(np.float64))
    sum1 = np.asarray(Gb.sum(axis=0)).ravel()
    sum0 = Gb.shape[0] - sum1
    F2 = F ** 2
    ssq1 = Gb.T @ F2
    ssq0 = F2.sum() - ssq1
    mean1 = Gb.T @ F
    mean0 = F.sum() - mean1
    with np.errstate(all="ignore"):
        mean1 = mean1 / (1e-8 + sum1)
        mean0 = mean0 / (1e-8 + sum0)
        var0 = (ssq0 - mean0 ** 2 * sum0) / (sum0 - 1)
        var1 = (ssq1 - mean1 ** 2 * sum1) / (sum1 - 1)
        varsum = var0 / sum0 + var1 / sum1
        dof = varsum ** 2 / (var0 / sum0 * (sum0 - 1) + var1 / sum1 * (sum1 - 1))
        f = mean1 - mean0
        t = f / np.sqrt(varsum)
    p = 2 * st.t.sf(np.abs(t), df=np.maximum(dof, 1))
    return f, t, p


def p_adjust_fdr(p):
    """stats::p.adjust(p, method="fdr") (Benjamini-Hochberg)."""
    p = np.asarray(p, dtype=np.float64)
    n = len(p)
    o = np.argsort(-p, kind="stable")
    q = np.minimum.accumulate(p[o] * n / np.arange(n, 0, -1))
    out = np.empty(n)
    out[o] = np.minimum(1.0, q)
    return out


def welch_ttests(gsetX, y):
    """Rfast::ttests(t(gsetX), ina=y+1) (R/plaid.R:429): Welch per row."""
    a = gsetX[:, y == 0]
    b = gsetX[:, y == 1]
    return st.ttest_ind(a, b, axis=1, equal_var=False).pvalue


def matrix_combine_p(plist, method="fisher"):
    """R/plaid.R:522-537."""
    if method in ("fisher", "sumlog"):
        chisq = -2 * sum(np.log(p) for p in plist)
        return st.chi2.sf(chisq, 2 * len(plist))
    if method in ("stouffer", "sumz"):
        zz = sum(st.norm.isf(p) for p in plist) / math.sqrt(len(plist))
        return st.norm.sf(zz)
    raise ValueError("Invalid method: " + method)


def plaid_test(X, rownames_x, y, G, rownames_g, gsetX, metap_method="fisher", tests=("one", "lm")):
    """R/plaid.R:392-474.  Returns dict of arrays in G-column order (no sorting); gsetX=None
    computes plaid(X, G) (:424-427)."""
    gg, ix, ig = align(rownames_g, rownames_x)            # :403 intersect(rownames(G), rownames(X))
    Xa = X[ig, :]
    Ga = sp.csc_matrix(G)[ix, :]
    m1 = np.asarray(Xa[:, y == 1].mean(axis=1)).ravel()   # :407
    m0 = np.asarray(Xa[:, y == 0].mean(axis=1)).ravel()   # :408
    fc = m1 - m0
    raw, eff = {}, []
    if "one" in tests:
        mean1, _, raw["p.one"] = matrix_onesample_ttest(fc, Ga)   # :413
        eff.append(mean1)
    if "two" in tests:
        diff, _, raw["p.two"] = matrix_twosample_ttest(fc, Ga)    # :419
        eff.append(diff)
    if "lm" in tests:
        if gsetX is None:
            gsetX = plaid(Xa, gg, Ga, gg)                          # :424-427
        gsetX = np.asarray(gsetX)
        raw["p.lm"] = welch_ttests(gsetX, y)                       # :429
        eff.append(gsetX[:, y == 1].mean(axis=1) - gsetX[:, y == 0].mean(axis=1))
    out = {}
    for k, p in raw.items():                                       # :441-446
        p = np.where(np.isnan(p), 1.0, p)
        out[k] = np.minimum(np.maximum(p, 1e-99), 1 - 1e-99)
    P = list(out.values())
    pmeta = matrix_combine_p(P, metap_method) if len(P) > 1 else P[0]
    out["p.meta"] = pmeta
    out["q.meta"] = p_adjust_fdr(pmeta)                            # :463
    out["gsetFC"] = np.mean(np.stack(eff, axis=1), axis=1)         # :453
    return out
