"""Host-side mirror of the reference's R API for the scoring hot path.

Same function names (dots -> underscores), argument meaning, defaults, dimnames and error
behaviour as R/plaid.R; the BODIES call the HIP library through the C ABI
(include/plaidhip.h).  There is no CPU compute path in here: gene-name alignment and
dimnames handling are host glue, everything numeric runs on the MI355X.
"""
from __future__ import annotations

import sys

import numpy as np
import scipy.sparse as sp

from ._lib import EUNSUPPORTED, PlaidHipError
from .engine import Context, default_context
from .gmt import GmtList, gmt2mat
from .matrix import NamedMatrix, as_named

INT_MAX = 2147483647  # .Machine$integer.max
_TIES = ("average", "min", "max", "first", "last", "dense", "random")   # what matrixStats::colRanks takes


def _message(txt: str):
    print(txt, file=sys.stderr)   # R message()


def _first_pos(names):
    pos = {}
    for k, nm in enumerate(names):
        pos.setdefault(nm, k)
    return pos


def aligned_pattern(X: NamedMatrix, matG: NamedMatrix):
    """Gene alignment + binarisation of R/plaid.R:65-73 without copying X:
    gg = intersect(rownames(X), rownames(matG)); G = 1*(matG[gg,] != 0), returned as a CSC
    pattern (Gp, Gi) whose row indices address X's rows.  None when nothing overlaps."""
    posx = _first_pos(X.rownames)
    G = sp.csc_matrix(matG.values)
    g2x = np.full(G.shape[0], -1, dtype=np.int64)
    seen = set()
    for k, nm in enumerate(matG.rownames):
        if nm in seen:
            continue                      # matG[gg,] picks the first row of that name
        seen.add(nm)
        r = posx.get(nm)
        if r is not None:
            g2x[k] = r
    if not np.any(g2x >= 0):
        return None
    new_idx = g2x[G.indices]
    keep = (new_idx >= 0) & (G.data != 0)
    m = G.shape[1]
    col = np.repeat(np.arange(m, dtype=np.int64), np.diff(G.indptr))
    counts = np.bincount(col[keep], minlength=m)
    Gp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=Gp[1:])
    if Gp[-1] > INT_MAX:
        raise PlaidHipError(EUNSUPPORTED, "membership matrix has more than 2^31-1 entries")
    return Gp.astype(np.int32), new_idx[keep].astype(np.int32)


def _auto_chunk(ncol_x: int) -> int:
    return int(np.round(0.8 * INT_MAX / max(ncol_x, 1)))     # R/plaid.R:103-104


def plaid(X, matG, stats=("mean", "sum"), chunk=None, normalize=True, ctx: Context | None = None):
    """plaid(), R/plaid.R:60-87.  Returns a NamedMatrix (sets x samples) or None with a
    message when no features overlap (:66-69)."""
    stats = stats if isinstance(stats, str) else stats[0]     # :62
    if stats not in ("mean", "sum"):
        raise ValueError("stats must be 'mean' or 'sum'")
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    Gp, Gi = pat
    ctx = ctx or default_context()
    g, n = X.shape
    m = matG.shape[1]
    auto = _auto_chunk(m)                                     # plaid passes chunk=NULL (:80)
    if n < auto:
        S = _crossprod_block(ctx, X, 0, n, Gp, Gi, stats, normalize)
    else:
        _message(f"[chunked_crossprod] chunked compute: chunk = {auto}")
        S = np.empty((m, n), dtype=np.float64, order="F")
        for j0 in range(0, n, auto):                          # :115-119
            j1 = min(n, j0 + auto)
            S[:, j0:j1] = _crossprod_block(ctx, X, j0, j1, Gp, Gi, stats, False)
        if normalize:
            S, _ = ctx.normalize_medians(S)                   # :83
    return NamedMatrix(S, matG.colnames, X.colnames)


def _crossprod_block(ctx, X: NamedMatrix, j0, j1, Gp, Gi, stats, normalize):
    if X.is_sparse:
        V = X.values[:, j0:j1] if (j0, j1) != (0, X.shape[1]) else X.values
        return ctx.plaid_csc(V.indptr, V.indices, V.data, X.shape[0], Gp, Gi, stats, normalize)
    return ctx.plaid_dense(X.values[:, j0:j1], Gp, Gi, stats, normalize)


def chunked_crossprod(x, y, chunk=None, ctx: Context | None = None):
    """chunked_crossprod(), R/plaid.R:100-123: t(x) %*% y, `x` genes x sets, `y` genes x samples with the same rows.
    A binary `x`, optionally column-scaled as plaid() builds it (:73-77), takes the scheduled membership kernels; an
    `x` whose stored values differ inside a column (weighted or signed sets) takes the general sparse kernel
    (plaidhip_crossprod_weighted_*).  Both run on the device; there is no host fallback."""
    x, y = as_named(x), as_named(y)
    if x.shape[0] != y.shape[0]:
        raise ValueError("non-conformable arguments")
    G = sp.csc_matrix(x.values)
    if not G.has_sorted_indices:
        G = G.sorted_indices()                                 # (a copy: the caller's matrix is not touched)
    m = G.shape[1]
    ctx = ctx or default_context()
    n = y.shape[1]
    if chunk is None or chunk < 0:
        chunk = _auto_chunk(m)
    scale = np.ones(m)
    nz = G.data != 0
    col = np.repeat(np.arange(m), np.diff(G.indptr))
    weighted = False
    if nz.any():
        vmin = np.full(m, np.inf)
        vmax = np.full(m, -np.inf)
        np.minimum.at(vmin, col[nz], G.data[nz])
        np.maximum.at(vmax, col[nz], G.data[nz])
        has = np.isfinite(vmin) & np.isfinite(vmax)
        # NaN / Inf weights or different values inside a column: not a (scaled) membership pattern
        weighted = bool(np.any(vmin[has] != vmax[has])) or not np.all(np.isfinite(G.data[nz]))
        if not weighted:
            scale[has] = vmin[has]
    if weighted:
        Wp, Wi, Wx = G.indptr.astype(np.int32), G.indices.astype(np.int32), G.data.astype(np.float64)
        g = x.shape[0]

        def block(j0, j1):
            if y.is_sparse:
                V = sp.csc_matrix(y.values[:, j0:j1])
                return ctx.crossprod_weighted(Wp, Wi, Wx, g, Yp=V.indptr, Yi=V.indices, Yx=V.data)
            return ctx.crossprod_weighted(Wp, Wi, Wx, g, Y=y.values[:, j0:j1])
    else:
        counts = np.bincount(col[nz], minlength=m)
        Gp = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(counts, out=Gp[1:])
        Gp, Gi = Gp.astype(np.int32), G.indices[nz].astype(np.int32)

        def block(j0, j1):
            return _crossprod_block(ctx, y, j0, j1, Gp, Gi, "sum", False)
    if n < chunk:
        S = block(0, n)
    else:
        _message(f"[chunked_crossprod] chunked compute: chunk = {chunk}")
        S = np.empty((m, n), dtype=np.float64, order="F")
        for j0 in range(0, n, chunk):
            j1 = min(n, j0 + chunk)
            S[:, j0:j1] = block(j0, j1)
    if not weighted:
        S *= scale[:, None]
    return NamedMatrix(S, x.colnames, y.colnames)


def normalize_medians(x, ignore_zero=None, ctx: Context | None = None):
    """normalize_medians(), R/plaid.R:554-575."""
    x = as_named(x)
    ctx = ctx or default_context()
    S, _ = ctx.normalize_medians(x.dense(), ignore_zero)
    return NamedMatrix(S, x.rownames, x.colnames)


def _check_ties(ties_method, allowed=_TIES):
    """ties.method is passed through like the reference does (R/plaid.R:614-617, 639-642); which values are legal depends on
    the function behind the branch: matrixStats::colRanks (all), base::rank (no "dense"), sparseMatrixStats::colRanks
    (max / average / min).  An illegal value raises like R's match.arg; "random" is legal in R and refused by the library
    (PLAIDHIP_EUNSUPPORTED: not a function of the input)."""
    if ties_method not in allowed:
        raise ValueError("'arg' should be one of " + ", ".join(f"\u2018{t}\u2019" for t in allowed))


def sparse_colranks(X, signed=False, ties_method="average", ctx: Context | None = None):
    """sparse_colranks(), R/plaid.R:631-650: ranks of the stored non-zeros per column; the
    sparsity pattern is kept, @x replaced."""
    _check_ties(ties_method, ("average", "first", "last", "random", "max", "min"))     # base::rank, :639-642
    X = as_named(X)
    V = sp.csc_matrix(X.values)
    ctx = ctx or default_context()
    rx = ctx.colranks_csc(V.indptr, V.data, ties_method, signed)
    R = sp.csc_matrix((rx, V.indices.copy(), V.indptr.copy()), shape=V.shape)
    return NamedMatrix(R, X.rownames, X.colnames)


def colranks(X, sparse=None, signed=False, keep_zero=False, ties_method="average",
             ctx: Context | None = None):
    """colranks(), R/plaid.R:589-623."""
    X = as_named(X)
    if sparse is None:
        sparse = X.is_sparse                                   # :595-596
    if sparse and keep_zero:
        return sparse_colranks(X, signed=signed, ties_method=ties_method, ctx=ctx)   # :600-601
    # the `sparse` ARGUMENT picks the function: sparseMatrixStats::colRanks (:603-608) or matrixStats::colRanks (:611-617)
    _check_ties(ties_method, ("max", "average", "min") if sparse else _TIES)
    ctx = ctx or default_context()
    if X.is_sparse and ties_method not in ("max", "average", "min"):
        X = NamedMatrix(X.values.toarray(), X.rownames, X.colnames)   # sparse = FALSE on a dgCMatrix: as.matrix(X), :617
    if X.is_sparse:
        # sparse without keep.zero: the reference's result is dense with the zeros ranked
        # (sparseMatrixStats::colRanks, :603-609) -- computed from the CSC arrays on the device
        V = X.values
        R = ctx.colranks_csc_dense(V.indptr, V.indices, V.data, X.shape[0], ties_method, signed)
    else:
        R = ctx.colranks_dense(X.values, ties_method, signed)                            # :612-618
    return NamedMatrix(R, X.rownames, X.colnames)


def replaid_sing(X, matG, ctx: Context | None = None):
    """replaid.sing(), R/plaid.R:213-219: min-ranks / nrow(X) - 0.5, then plaid(normalize=FALSE)."""
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    if X.is_sparse:                                        # the CSC slots go to the device: no dense X on the host
        V = X.values
        S = ctx.sing_csc(V.indptr, V.indices, V.data, X.shape[0], pat[0], pat[1])
    else:
        S = ctx.sing_dense(X.values, pat[0], pat[1])
    return NamedMatrix(S, matG.colnames, X.colnames)


def replaid_ssgsea(X, matG, alpha=0, ctx: Context | None = None):
    """replaid.ssgsea(), R/plaid.R:244-255: average ranks (non-zeros only for sparse X, :245 ->
    :600-601), ^(1+alpha), / global max - 0.5, then plaid(stats="mean", normalize=TRUE)."""
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    if X.is_sparse:
        V = X.values
        S = ctx.ssgsea_csc(V.indptr, V.indices, V.data, X.shape[0], pat[0], pat[1], float(alpha))
    else:
        S = ctx.ssgsea_dense(X.values, pat[0], pat[1], float(alpha))
    return NamedMatrix(S, matG.colnames, X.colnames)


def _set_sizes_unaligned(matG: NamedMatrix) -> np.ndarray:
    """Matrix::colSums(matG != 0) of the matrix as given (R/plaid.R:280 does not re-align)."""
    G = sp.csc_matrix(matG.values)
    col = np.repeat(np.arange(G.shape[1]), np.diff(G.indptr))
    return np.bincount(col[G.data != 0], minlength=G.shape[1]).astype(np.float64)


def replaid_ucell(X, matG, rmax=1500, ctx: Context | None = None):
    """replaid.ucell(), R/plaid.R:276-282."""
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    S = ctx.ucell(X.values, pat[0], pat[1], _set_sizes_unaligned(matG), float(rmax))
    return NamedMatrix(S, matG.colnames, X.colnames)


def replaid_aucell(X, matG, aucMaxRank=None, ctx: Context | None = None):
    """replaid.aucell(), R/plaid.R:304-309; aucMaxRank defaults to ceiling(0.05 * nrow(X))."""
    X, matG = as_named(X), as_named(matG)
    if aucMaxRank is None:
        aucMaxRank = int(np.ceil(0.05 * X.shape[0]))
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    S = ctx.aucell(X.values, pat[0], pat[1], float(aucMaxRank))
    return NamedMatrix(S, matG.colnames, X.colnames)


def replaid_scse(X, matG, removeLog2=None, scoreMean=False, ctx: Context | None = None):
    """replaid.scse(), R/plaid.R:155-190."""
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    S = ctx.scse(X.values, pat[0], pat[1], removeLog2, scoreMean)
    if ctx.last_scse_removed_log2:        # R/plaid.R:163-164 (removeLog2 = NULL is decided on the device, :160-161)
        _message("[replaid.scse] Converting data to linear scale (removing log2)...")
    return NamedMatrix(S, matG.colnames, X.colnames)


def replaid_gsva(X, matG, tau=0, rowtf="z", ctx: Context | None = None):
    """replaid.gsva(), R/plaid.R:338-363 (row z-transform variant; the result of plaid() on the rank matrix
    carries the dimnames of matG / X)."""
    rowtf = rowtf if isinstance(rowtf, str) else rowtf[0]
    X, matG = as_named(X), as_named(matG)
    pat = aligned_pattern(X, matG)
    if pat is None:
        _message("[plaid] ERROR. No overlapping features.")
        return None
    ctx = ctx or default_context()
    Xv = X.values.toarray() if sp.issparse(X.values) else X.values
    S = ctx.gsva(Xv, pat[0], pat[1], float(tau), rowtf)
    return NamedMatrix(S, matG.colnames, X.colnames)


_TEST_BITS = {"one": 1, "two": 2, "lm": 4}


def plaid_test(X, y, G, gsetX=None, tests=("one", "two", "lm"), metap_method="fisher", sort_by="p.meta",
               ctx: Context | None = None):
    """plaid.test(), R/plaid.R:392-474: one-/two-sample t-tests of the logFC inside each set plus a Welch test
    of the single-sample scores between the two groups, combined by Fisher or Stouffer, BH-adjusted.
    The statistics are reduced on the device; with gsetX=None the scores plaid(X, G) never leave it.
    Returns a NamedMatrix (sets x [gsetFC, p.<test>..., p.meta, q.meta]) ordered by `sort_by` (:469-471)."""
    y = np.asarray(y)
    if not np.all(np.isin(np.unique(y), (0, 1))):
        raise ValueError("elements of y must be 0 or 1")                      # :394
    if isinstance(G, (GmtList, dict)) or (isinstance(G, tuple) and len(G) == 2):
        _message("[plaid.test] converting gmt to sparse matrix...")           # :396-397
        G = gmt2mat(G)
    X, G = as_named(X), as_named(G)
    tests = [tests] if isinstance(tests, str) else list(tests)
    bits = 0
    for t in tests:
        if t not in _TEST_BITS:
            raise ValueError(f"unknown test {t!r}")
        bits |= _TEST_BITS[t]
    if metap_method in ("fisher", "sumlog"):
        mm = 0
    elif metap_method in ("stouffer", "sumz"):
        mm = 1
    else:
        raise ValueError("Invalid method: " + str(metap_method))              # :533
    # gg <- intersect(rownames(G), rownames(X)); X <- X[gg,]; G <- G[gg,]    (:403-405)
    posx = _first_pos(X.rownames)
    seen, grow, xrow = set(), [], []
    for k, nm in enumerate(G.rownames):
        if nm in seen:
            continue
        seen.add(nm)
        r = posx.get(nm)
        if r is not None:
            grow.append(k)
            xrow.append(r)
    Gs = sp.csc_matrix(G.values)[grow, :].tocsc()
    Gs.sort_indices()
    keep = Gs.data != 0
    counts = np.bincount(np.repeat(np.arange(Gs.shape[1]), np.diff(Gs.indptr))[keep], minlength=Gs.shape[1])
    Gp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    Gi = Gs.indices[keep].astype(np.int32)
    Xv = X.values
    Xs = np.asfortranarray(Xv[xrow, :].toarray() if sp.issparse(Xv) else np.asarray(Xv)[xrow, :], dtype=np.float64)
    sx = None
    if gsetX is not None:
        gx = as_named(gsetX)
        if gx.rownames is not None and list(gx.rownames) != list(G.colnames):
            pos = _first_pos(gx.rownames)
            sx = np.asarray(gx.values)[[pos[nm] for nm in G.colnames], :]
        else:
            sx = np.asarray(gx.values)
    ctx = ctx or default_context()
    out = ctx.plaid_test(Xs, y.astype(np.int32), Gp, Gi, sx, bits, mm)
    cols, names = [0], ["gsetFC"]
    for t, c in (("one", 1), ("two", 2), ("lm", 3)):
        if t in tests:
            cols.append(c)
            names.append("p." + t)
    cols += [4, 5]
    names += ["p.meta", "q.meta"]
    res = out[:, cols]
    rn = list(G.colnames)
    if sort_by in names:
        o = np.argsort(res[:, names.index(sort_by)], kind="stable")       # order()
        res = res[o, :]
        rn = [rn[k] for k in o]
    return NamedMatrix(res, rn, names)
