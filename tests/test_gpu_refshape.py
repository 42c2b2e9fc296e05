"""The HIP path on gene-set collections with the SHAPE of the reference's own benchmark collection -- `playdata::GSETxGENE`,
61,459 sets (experiments/benchmark/benchmark-plaid.R:18-35; published timings benchmark-pbmc3k@p14.csv:133,
benchmark-brca@p14.csv:133): set sizes 3 ... 5,000 and one set with EVERY gene, Zipf gene popularity (hub genes in 10-16 %
of all sets), parent / child / duplicate sets (plaid_amd.synth.geneset_csc_real).  Every kernel that consumes a
prepared collection is compared with the oracle on it: the tile planner (a tile is as long as its longest lane), the
scatter plan's per-gene id lists (a hub gene fills ~25 segments per chunk), the u16 quad kernel's 32-bit sums (an all-genes
set sums 2 * rank over every gene), the fixed-point scatter accumulators (kbits of a 20,000-gene set), the medians over
61k scores, the multi-shard engine.  `pytest -m gpu`."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-9


def close(a, b):
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=ATOL)


def _po():
    from oracle import plaid_oracle
    return plaid_oracle


_CACHE = {}


def _collection(g, m, hubs="scattered"):
    from plaid_amd import synth as sy
    key = (g, m, hubs)
    if key not in _CACHE:
        Gp, Gi = sy.geneset_csc_real(g, m, hubs=hubs)
        _CACHE[key] = (Gp, Gi, sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m)), [str(k) for k in range(g)])
    return _CACHE[key]


# the two shapes the reference published timings for (genes x sets), and shapes around the planners' slice boundaries:
# 8,000 genes (the u16 kernel does not apply), 20,448 (the largest one-slice column), 25,000 (two one-column slices,
# three pair slices: the all-genes set spans all of them)
SHAPES = [(12010, 61459), (17713, 61510)]
SMALL = [(8000, 3000), (20448, 5000), (25000, 4000), (300, 700)]


def test_generator_has_the_reference_shape():
    Gp, Gi, G, rn = _collection(12010, 61459)
    k = np.diff(Gp)
    assert k[0] == 12010 and k.min() >= 1 and (k >= 3000).sum() >= 3 and np.all(np.diff(k) <= 0)
    f = np.bincount(Gi, minlength=12010) / 61459.0
    assert (f >= 0.10).sum() >= 5 and np.median(f) < 0.01 and f.min() > 0
    assert 4_000_000 < len(Gi) < 7_000_000
    for j in (0, 1, 100, 61458):
        assert np.all(np.diff(Gi[Gp[j]:Gp[j + 1]]) > 0)


@pytest.mark.parametrize("g,m", SHAPES + SMALL)
@pytest.mark.parametrize("kernel", ["auto", "single"])
def test_plaid_dense_on_the_reference_shaped_collection(pinned_ctx, g, m, kernel):
    """plaid(X, matG) R/plaid.R:60-87, dense X: pair kernel (default) and one-column kernel, raw and median-normalised"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    Gp, Gi, G, rn = _collection(g, m)
    n = 13
    X = sy.dense_columns(g, 0, n)
    ctx = pinned_ctx(spmm_dense_kernel=kernel)
    close(ctx.plaid_dense(X, Gp, Gi, "mean", False), c_oracle.plaid_dense(X, Gp, Gi, "mean", False))
    close(ctx.plaid_dense(X, Gp, Gi, "mean", True), c_oracle.plaid_dense(X, Gp, Gi, "mean", True))
    close(ctx.plaid_dense(X, Gp, Gi, "sum", False), c_oracle.plaid_dense(X, Gp, Gi, "sum", False))


@pytest.mark.parametrize("hubs", ["front", "scattered"])
def test_plaid_dense_hub_genes_in_the_first_rows(hip_ctx, hubs):
    """a matrix straight from gmt2mat has its rows ordered by decreasing frequency (R/gmt-utils.R:31,62): every hub gene
    then falls into the FIRST gene slice and onto a few LDS slots' neighbourhood"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    g, m = 17713, 20000
    Gp, Gi, G, rn = _collection(g, m, hubs)
    X = sy.dense_columns(g, 0, 6)
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", True), c_oracle.plaid_dense(X, Gp, Gi, "mean", True))
    Xp, Xi, Xx = sy.sparse_columns(g, 0, 6)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, 6))
    close(hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True), _po().plaid(Xs, rn, G, rn))


@pytest.mark.parametrize("g,m", SHAPES + SMALL)
@pytest.mark.parametrize("mode", ["scatter", "gather", "auto"])
def test_plaid_csc_on_the_reference_shaped_collection(pinned_ctx, g, m, mode):
    """plaid() on a dgCMatrix (R/plaid.R:107, sparse branch): scatter kernel (fixed-point and fp64 accumulators), gather
    kernel, and the device-side choice; a hub gene's id list spans many 128-id segments per chunk of sets"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    Gp, Gi, G, rn = _collection(g, m)
    n = 11
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=0.06)
    exp_raw = c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi, "mean", threads=8)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    for fixed in ("on", "off"):
        ctx = pinned_ctx(spmm_sparse_kernel=mode, scatter_fixed=fixed)
        close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", False), exp_raw)
    close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True), _po().plaid(Xs, rn, G, rn))
    close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "sum", False),
          c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi, "sum", threads=8))


@pytest.mark.parametrize("g,m", SHAPES + SMALL)
def test_sing_on_the_reference_shaped_collection(pinned_ctx, g, m):
    """replaid.sing (R/plaid.R:213-219): min ranks, the u16 quad kernel's 32-bit integer sums (the all-genes set sums
    2 * rank over EVERY gene: g (g + 1) < 2^32) -- bit-identical to the fp64 kernels, and equal to the oracle"""
    from plaid_amd import synth as sy
    Gp, Gi, G, rn = _collection(g, m)
    n = 9
    X = sy.dense_columns(g, 0, n, tied=True)
    a = pinned_ctx(ranks_f32=2).sing_dense(X, Gp, Gi)
    b = pinned_ctx(ranks_f32=0).sing_dense(X, Gp, Gi)
    assert np.array_equal(a, b)
    close(a, _po().replaid_sing(X, rn, G, rn))
    # the all-genes set on tie-free columns: the ranks are a permutation of 1..g, so the score is (g + 1) / (2 g) - 0.5
    # in every sample -- the largest integer sum the u16 kernel forms, g (g + 1)
    a0 = pinned_ctx(ranks_f32=2).sing_dense(sy.dense_columns(g, 0, 5), Gp, Gi)
    assert np.diff(Gp)[0] == g and np.allclose(a0[0], (g + 1) / (2.0 * g) - 0.5, rtol=0, atol=1e-12)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    close(pinned_ctx().sing_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi), _po().replaid_sing(Xs, rn, G, rn))


@pytest.mark.parametrize("g,m", SHAPES + SMALL[:3])
def test_ssgsea_on_the_reference_shaped_collection(pinned_ctx, g, m):
    """replaid.ssgsea(alpha = 0.25) (R/plaid.R:244-255) dense and dgCMatrix: rank weights up to g^1.25 summed over sets of up
    to g genes -- the scatter kernel's fixed-point headroom is kbits(largest set) + bits(max weight) <= 63"""
    from plaid_amd import synth as sy
    Gp, Gi, G, rn = _collection(g, m)
    n = 7
    X = sy.dense_columns(g, 0, n, tied=True)
    close(pinned_ctx().ssgsea_dense(X, Gp, Gi, 0.25), _po().replaid_ssgsea(X, rn, G, rn, alpha=0.25))
    close(pinned_ctx().ssgsea_dense(X, Gp, Gi, 0.0), _po().replaid_ssgsea(X, rn, G, rn, alpha=0.0))
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=0.08)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    exp = _po().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25)
    for mode in ("scatter", "gather"):
        for fixed in ("on", "off"):
            ctx = pinned_ctx(spmm_sparse_kernel=mode, scatter_fixed=fixed)
            close(ctx.ssgsea_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, 0.25), exp)


def test_normalize_medians_of_61k_scores_with_many_ties(hip_ctx):
    """normalize_medians (R/plaid.R:554-575) at m = 61,459: duplicate sets give exactly equal scores (ties at the median),
    both ignore.zero branches"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    g, m = 12010, 61459
    Gp, Gi, G, rn = _collection(g, m)
    X = sy.dense_columns(g, 0, 5)
    raw = c_oracle.plaid_dense(X, Gp, Gi, "mean", False)
    for iz in (None, False, True):
        S = np.asfortranarray(raw.copy())
        if iz is not False:
            S[::7, :] = 0.0
        exp, med_o = c_oracle.normalize_medians(S, iz)
        got, med = hip_ctx.normalize_medians(S, iz)
        close(got, exp)
        close(med, med_o)


@pytest.mark.parametrize("nshards", [2, 3])
def test_multi_shard_engine_on_the_reference_shaped_collection(hip_ctx, nshards):
    """plaidhip_*_multi's engine (thread per shard) with the 61,459-set collection: dense plaid / sing / ssgsea equal the
    one-context results bit for bit, the sparse route meets the oracle"""
    import ctypes as C
    from plaid_amd import synth as sy
    from plaid_amd._lib import load
    lib = load()
    fn = lib.plaidhip_debug_sharded_on_one_device
    vp = C.c_void_p
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int, C.c_int,
                   C.c_double, vp]
    g, m = 12010, 61459
    Gp, Gi, G, rn = _collection(g, m)
    n = 2 * nshards + 1
    X = sy.dense_columns(g, 0, n, tied=True)

    def run(method, Xf=None, Xs=None, alpha=0.0):
        S = np.full((m, n), np.nan, order="F")
        if Xs is None:
            rc = fn(0, nshards, -1, method, None, None, Xf.ctypes.data, g, n, Gp.ctypes.data, Gi.ctypes.data, m, 0, 1, alpha,
                    S.ctypes.data)
        else:
            p_, i_, x_ = (np.ascontiguousarray(Xs.indptr, dtype=np.int32), np.ascontiguousarray(Xs.indices, dtype=np.int32),
                          np.ascontiguousarray(Xs.data, dtype=np.float64))
            rc = fn(0, nshards, -1, method, p_.ctypes.data, i_.ctypes.data, x_.ctypes.data, g, n, Gp.ctypes.data, Gi.ctypes.data,
                    m, 0, 1, alpha, S.ctypes.data)
        assert rc == 0, lib.plaidhip_last_error_string()
        return S

    Xf = np.asfortranarray(X)
    assert np.array_equal(run(0, Xf), hip_ctx.plaid_dense(X, Gp, Gi, "mean", True))
    assert np.array_equal(run(1, Xf), hip_ctx.sing_dense(X, Gp, Gi))
    assert np.array_equal(run(2, Xf, alpha=0.25), hip_ctx.ssgsea_dense(X, Gp, Gi, 0.25))
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    close(run(0, Xs=Xs), _po().plaid(Xs, rn, G, rn))
    close(run(2, Xs=Xs, alpha=0.25), _po().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))


def test_geneset_create_at_the_reference_size_stays_within_its_budget(hip_ctx):
    """plaidhip_geneset_create on 61,459 real-shaped sets: seconds, not minutes (the reference's gmt2mat alone takes 50.9 s
    for 50k sets, experiments/benchmark/benchmark-plaid.R:42), and a padded index volume within 1.5x of the memberships"""
    import time
    g, m = 12010, 61459
    Gp, Gi, G, rn = _collection(g, m)
    t0 = time.time()
    gs = hip_ctx.geneset(g, Gp, Gi)
    dt = time.time() - t0
    info = gs.info()
    gs.close()
    z = len(Gi)
    assert info["z"] == z
    assert dt < 20.0, dt
    eff1, eff2 = z / info["padded_slots"], z / info["padded_slots_pair"]          # slot efficiency: one-column / pair plan
    assert eff1 > 0.70 and eff2 > 0.64, (eff1, eff2)


def test_rank_weights_keep_fixed_point_sums_next_to_an_all_genes_set():
    """replaid.ssgsea on a dgCMatrix with the reference-shaped collection: (largest set size) x max weight would leave the u64
    grid 35 fraction bits (a 12,010-gene set: kbits = 14) and the sums would go to fp64 atomics, whose last bits depend on
    the arrival order; the largest COLUMN sum bounds every score as well (all weights >= 0) and leaves 41: the scores are
    bit-identical between runs and between the kernel's two item orders, and within 1e-12 of the fp64-atomic sums.  A
    collection the first bound serves keeps its grid (same bits as with the column-sum pass switched off by a small
    collection: the synthetic one, kbits = 9, takes no column-sum pass at all)"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    g, m, n, alpha = 12010, 30000, 900, 0.25
    Gp, Gi = sy.geneset_csc_real(g, m)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=0.07)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    outs = {}
    with torch.cuda.stream(stream):
        dp, di, dx = (torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (Xp.astype(np.int32), Xi.astype(np.int32), Xx))
        Rx = torch.empty_like(dx)
        colmax = torch.zeros(n, dtype=torch.float64, device=dev)
        gmax = torch.zeros(1, dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.dev_colranks_csc(dp.data_ptr(), dx.data_ptr(), n, int(np.diff(Xp).max()), Rx.data_ptr(), "average", False, 1.0 + alpha,
                             colmax.data_ptr())
        ctx.dev_max(colmax.data_ptr(), n, gmax.data_ptr())
        ctx.set_option("spmm_sparse_kernel", "scatter")
        for key, fixed, order in (("fx_chunk", "on", "chunk"), ("fx_chunk2", "on", "chunk"), ("fx_col", "on", "column"),
                                  ("f64", "off", "chunk")):
            ctx.set_option("scatter_fixed", fixed)
            ctx.set_option("scatter_order", order)
            S = torch.empty((n, m), dtype=torch.float64, device=dev)
            ctx.dev_spmm_csc_ranks(gs, dp.data_ptr(), di.data_ptr(), Rx.data_ptr(), n, S.data_ptr(), m, gmax.data_ptr(), "mean",
                                   1.0, -0.5, flags.data_ptr(), nnz=len(Xx))
            outs[key] = S
    torch.cuda.synchronize()
    assert torch.equal(outs["fx_chunk"], outs["fx_chunk2"]) and torch.equal(outs["fx_chunk"], outs["fx_col"])
    assert float((outs["f64"] - outs["fx_chunk"]).abs().max()) < 1e-12
    # against the oracle on a few cells
    from oracle import c_oracle
    cols = [0, 1, n - 1]
    Rh = Rx.cpu().numpy() / float(gmax[0])
    sub_p = np.zeros(len(cols) + 1, np.int32)
    sub_i, sub_x = [], []
    for k, c in enumerate(cols):
        sub_i.append(Xi[Xp[c]:Xp[c + 1]]); sub_x.append(Rh[Xp[c]:Xp[c + 1]])
        sub_p[k + 1] = sub_p[k] + Xp[c + 1] - Xp[c]
    exp = c_oracle.crossprod_csc(sub_p, np.concatenate(sub_i), np.concatenate(sub_x), g, Gp, Gi, "mean", threads=4) - \
        0.5 * (np.diff(Gp) * (1.0 / (1e-8 + np.diff(Gp))))[:, None]
    np.testing.assert_allclose(outs["fx_chunk"].cpu().numpy()[cols].T, exp, rtol=1e-9, atol=1e-12)
    gs.close()
    ctx.close()
