"""Synthetic genes x samples x genesets inputs (SURVEY.md section 8d / BASELINE.md section 3).

One generator feeds the HIP path, the tests and the CPU baseline with identical
bytes.  Seeds are fixed (`X` 20250614, `G` 20250615); columns are generated in
independent 256-column blocks so any column range (a shard of a rank, the bounded
CPU-baseline sample) can be reproduced without generating the whole matrix.

Shapes follow the reference's conventions: expression is genes x samples,
column-major (`order="F"`, R layout); membership is a genes x sets 0/1 CSC matrix
like `gmt2mat()` returns (R/gmt-utils.R:19-66): sets ordered by decreasing size.
"""
from __future__ import annotations

import numpy as np

SEED_X = 20250614
SEED_G = 20250615
BLOCK = 256


def geneset_csc(g: int, m: int, seed: int = SEED_G, kmin: int = 15, kmax: int = 500,
                sort_by_size: bool = True):
    """m gene sets over g genes; size k_j = round(exp(U(ln kmin, ln kmax))) (mean about
    138 for 15..500, cf. inst/extdata/hallmarks.gmt sizes 32..200); members uniform
    without replacement; CSC with sorted row indices.  Returns (Gp int32[m+1], Gi int32[z])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kmax = min(kmax, g)
    kmin = min(kmin, kmax)
    sizes = np.rint(np.exp(rng.uniform(np.log(kmin), np.log(kmax), size=m))).astype(np.int64)
    sizes = np.clip(sizes, 1, g)
    if sort_by_size:
        sizes = -np.sort(-sizes, kind="stable")   # gmt2mat: decreasing size
    Gp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(sizes, out=Gp[1:])
    Gi = np.empty(int(Gp[-1]), dtype=np.int32)
    for j in range(m):
        k = int(sizes[j])
        # Floyd-free: permutation prefix is fine at these sizes
        Gi[Gp[j]:Gp[j + 1]] = np.sort(rng.choice(g, size=k, replace=False))
    return Gp.astype(np.int32), Gi


def geneset_csc_real(g: int, m: int, seed: int = SEED_G + 7, hubs: str = "scattered", all_genes_set: bool = True,
                     kmin: int = 5, kmax: int = 5000):
    """A gene-set collection with the SHAPE of the one the reference benchmarks with -- `playdata::GSETxGENE`, 61,459 sets
    over the genes of the data set (experiments/benchmark/benchmark-plaid.R:18-35; the package is not in the tree, so
    its shape is restated, not its content):

      * set sizes log-normal (median 40, sigma 1.2) clipped to [kmin, kmax] = 5 ... 5,000 genes -- a long right tail, and
        at least three sets beyond 3,000 genes once m >= 1,000;
      * ONE set that holds every gene (`all_genes_set`), the largest the planner can meet;
      * Zipf-like gene popularity, p(gene of popularity rank r) ~ (r + 20)^-0.7: a handful of hub genes sit in 10-25 % of
        all sets, the median gene in ~0.5 %;
      * duplicate-heavy overlaps: 15 % of the sets are children of an earlier set (90 % of the parent's genes + 10 % new
        ones), 1 % are exact copies under another name -- what curated collections (GO parents / children, the same
        pathway from several sources) look like;
      * columns ordered by decreasing size like gmt2mat() output (R/gmt-utils.R:25).

    hubs: "front" keeps popularity rank = row index (gmt2mat orders ROWS by decreasing frequency, R/gmt-utils.R:31,62 --
    the layout of a matrix straight from gmt2mat); "scattered" (default) permutes the genes, which is what plaid() sees
    after re-indexing the pattern into X's row space (R/plaid.R:65-72: the order of rownames(X)).
    Returns (Gp int32[m+1], Gi int32[z] sorted inside a column)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kmax = min(kmax, g)
    kmin = max(1, min(kmin, kmax))
    sizes = np.rint(np.exp(rng.normal(np.log(40.0), 1.2, size=m))).astype(np.int64)
    sizes = np.clip(sizes, kmin, kmax)
    if m >= 1000:
        sizes[:3] = np.maximum(sizes[:3], np.minimum(kmax, 3000 + np.arange(3) * 800))
    w = (np.arange(g, dtype=np.float64) + 20.0) ** -0.7
    cw = np.cumsum(w / w.sum())
    sets = [None] * m
    # weighted draws WITHOUT replacement: oversampled draws with replacement, de-duplicated in drawing order; sets that
    # take a large part of the genes (the draws would mostly repeat) use the exponential-keys method instead
    for j in range(m):
        k = int(sizes[j])
        r = rng.random()
        if j >= 10 and r < 0.01:                                   # exact copy of an earlier set
            sets[j] = sets[int(rng.integers(0, j))]
            continue
        if j >= 10 and r < 0.16:                                   # child: 90 % of the parent + new genes
            par = sets[int(rng.integers(0, j))]
            keep = par[rng.random(len(par)) < 0.9]
            extra = np.searchsorted(cw, rng.random(max(1, len(par) // 10)))
            sets[j] = np.unique(np.concatenate([keep, np.minimum(extra, g - 1)])).astype(np.int32)
            continue
        if k * 6 > g:
            keys = rng.exponential(size=g) / w
            sets[j] = np.sort(np.argpartition(keys, k - 1)[:k]).astype(np.int32)
            continue
        got = np.zeros(0, dtype=np.int64)
        while len(got) < k:
            d = np.minimum(np.searchsorted(cw, rng.random(int(1.4 * (k - len(got))) + 8)), g - 1)
            allv = np.concatenate([got, d])
            _, first = np.unique(allv, return_index=True)
            got = allv[np.sort(first)]
        sets[j] = np.sort(got[:k]).astype(np.int32)
    if all_genes_set and m > 0:
        sets[0] = np.arange(g, dtype=np.int32)
    if hubs == "scattered":
        perm = rng.permutation(g).astype(np.int32)
        sets = [np.sort(perm[s_]) for s_ in sets]
    elif hubs != "front":
        raise ValueError("hubs: 'scattered' or 'front'")
    order = np.argsort(-np.array([len(s_) for s_ in sets]), kind="stable")   # gmt2mat: decreasing size
    sets = [sets[j] for j in order]
    Gp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum([len(s_) for s_ in sets], out=Gp[1:])
    assert Gp[-1] < 2**31 - 1
    Gi = np.concatenate(sets).astype(np.int32) if m else np.zeros(0, np.int32)
    return Gp.astype(np.int32), Gi


def _block_rng(seed: int, block: int):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, block])))


def dense_columns(g: int, j0: int, j1: int, seed: int = SEED_X, tied: bool = False):
    """Columns [j0, j1) of the dense expression matrix: N(8, 2^2) doubles (tie-free) or,
    with `tied`, rounded to one decimal (heavy ties for the rank kernels).
    Returns a Fortran-ordered (g, j1-j0) float64 array."""
    out = np.empty((g, j1 - j0), dtype=np.float64, order="F")
    b0, b1 = j0 // BLOCK, (j1 - 1) // BLOCK if j1 > j0 else j0 // BLOCK
    for b in range(b0, b1 + 1):
        rng = _block_rng(seed, b)
        blk = rng.normal(8.0, 2.0, size=(BLOCK, g))      # row = one sample column
        lo, hi = max(j0, b * BLOCK), min(j1, (b + 1) * BLOCK)
        out[:, lo - j0:hi - j0] = blk[lo - b * BLOCK:hi - b * BLOCK].T
    if tied:
        np.round(out, 1, out=out)
    return out


def sparse_columns(g: int, j0: int, j1: int, seed: int = SEED_X + 1, density: float = 0.05,
                   levels: int = 50):
    """Columns [j0, j1) of the sparse (about 95 % zero) expression matrix as CSC arrays
    (Xp int32, Xi int32 sorted, Xx float64).  Per cell Binomial(g, density) non-zero rows,
    uniform; values log1p(count / size_factor) with counts from a geometric-like law so
    that a column has about `levels` distinct values (the reference fixture: 52 distinct
    values in 1,352 non-zeros)."""
    ps, idx, val = [0], [], []
    for j in range(j0, j1):
        rng = _block_rng(seed, j)
        nnz = int(rng.binomial(g, density))
        rows = np.sort(rng.choice(g, size=nnz, replace=False)).astype(np.int32)
        counts = np.minimum(rng.geometric(0.12, size=nnz), levels).astype(np.float64)
        sf = 0.5 + rng.random()                            # per-cell size factor
        idx.append(rows)
        val.append(np.log1p(counts / sf))
        ps.append(ps[-1] + nnz)
    Xi = np.concatenate(idx) if idx else np.zeros(0, np.int32)
    Xx = np.concatenate(val) if val else np.zeros(0, np.float64)
    return np.asarray(ps, dtype=np.int64), Xi, Xx


def device_sparse_cells(torch, dev, g, n, seed, density=0.05, levels=50):
    """n cells of the sparse workload generated ON the device (~5 % stored values per cell, values
    log1p(count / size factor) with ~50 levels per cell), as CSC tensors.  Returns (p int32[n+1], i int32, x float64,
    nnz, longest column)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    ps, idx, val = [torch.zeros(1, dtype=torch.int64, device=dev)], [], []
    B = 4096
    for j0 in range(0, n, B):
        b = min(B, n - j0)
        mask = torch.rand((b, g), device=dev, generator=gen) < density
        cnt = mask.sum(dim=1)
        nz = mask.nonzero(as_tuple=False)                  # sorted by cell, then gene
        k = torch.empty(nz.shape[0], device=dev, dtype=torch.float64).geometric_(0.12, generator=gen).clamp_(max=levels)
        sf = 0.5 + torch.rand(b, device=dev, dtype=torch.float64, generator=gen)
        val.append(torch.log1p(k / sf[nz[:, 0]]))
        idx.append(nz[:, 1].to(torch.int32))
        ps.append(cnt)
        del mask, nz, k
    cnt = torch.cat(ps)
    p = torch.cumsum(cnt, 0)
    nnz = int(p[-1].item())
    assert nnz < 2**31 - 1, "more than 2^31-1 stored values: 32-bit dgCMatrix slots"
    return p.to(torch.int32), torch.cat(idx), torch.cat(val), nnz, int(cnt.max().item())
