"""Parity of the HIP path (through the C ABI) with the CPU oracle -- `pytest -m gpu`.

Bars (BASELINE.json north_star): ranks bit-exact; float scores within 1e-5 relative.
The score tolerance used here is |a-b| <= RTOL*|b| + ATOL with RTOL = 1e-5 and
ATOL = 1e-9 (fp64 accumulation leaves ~1e-13, the ATOL only covers scores that cancel to
~0 after the -0.5 centring of R/plaid.R:216,251).
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-9


def close(a, b):
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=ATOL)


def _oracle():
    from oracle import plaid_oracle
    return plaid_oracle


# ---------------------------------------------------------------- golden: synthetic cases
@pytest.mark.parametrize("tm", ["average", "min", "max"])
@pytest.mark.parametrize("signed", [False, True])
def test_colranks_dense_golden(hip_ctx, synth, tm, signed):
    R = hip_ctx.colranks_dense(synth["rank_X"], tm, signed)
    exp = synth[f"rank_signed_{tm}" if signed else f"rank_{tm}"]
    assert np.array_equal(R, exp)          # bit-exact (half-)integers


@pytest.mark.parametrize("tm", ["average", "min", "max"])
@pytest.mark.parametrize("signed", [False, True])
def test_sparse_colranks_golden(hip_ctx, synth, tm, signed):
    R = hip_ctx.colranks_csc(synth["rank_csc_p"], synth["rank_csc_x"], tm, signed)
    exp = synth[f"rank_csc_signed_{tm}" if signed else f"rank_csc_{tm}"]
    assert np.array_equal(R, exp)


def test_crossprod_golden(hip_ctx, synth):
    X, Gp, Gi = synth["cp_X"], synth["cp_Gp"], synth["cp_Gi"]
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", False), synth["cp_mean_raw"])
    close(hip_ctx.plaid_dense(X, Gp, Gi, "sum", False), synth["cp_sum_raw"])
    S = hip_ctx.plaid_dense(X, Gp, Gi, "mean", False)
    assert np.all(S[3, :] == 0.0)          # empty set: exactly 0 (R/plaid.R:75-76: 0 * 1e8)


def test_plaid_normalised_golden(hip_ctx, synth):
    X, Gp, Gi = synth["cp_X"], synth["cp_Gp"], synth["cp_Gi"]
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", True), synth["cp_mean_norm"])      # ignore.zero branch
    close(hip_ctx.plaid_dense(X - 8.0, Gp, Gi, "mean", True), synth["cp_neg_norm"])  # negatives: plain medians
    # drop the empty set -> no exact zeros -> plain medians
    keep = [j for j in range(len(Gp) - 1) if j != 3]
    sizes = np.diff(Gp)
    Gp2 = np.concatenate([[0], np.cumsum(sizes[keep])]).astype(np.int32)
    Gi2 = np.concatenate([Gi[Gp[j]:Gp[j + 1]] for j in keep]).astype(np.int32)
    close(hip_ctx.plaid_dense(X, Gp2, Gi2, "mean", True), synth["cp_nozero_norm"])


def test_plaid_csc_equals_dense(hip_ctx, synth):
    X, Gp, Gi = synth["cp_X"].copy(), synth["cp_Gp"], synth["cp_Gi"]
    X[np.random.default_rng(3).random(X.shape) < 0.8] = 0.0
    Xs = sp.csc_matrix(X)
    a = hip_ctx.plaid_dense(X, Gp, Gi, "mean", True)
    b = hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, X.shape[0], Gp, Gi, "mean", True)
    close(b, a)
    rn = [f"g{k}" for k in range(X.shape[0])]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(X.shape[0], len(Gp) - 1))
    close(b, _oracle().plaid(Xs, rn, G, rn))


def test_normalize_medians_golden(hip_ctx, synth):
    close(hip_ctx.normalize_medians(synth["nm_S"])[0], synth["nm_auto"])
    close(hip_ctx.normalize_medians(synth["nm_S"], True)[0], synth["nm_true"])
    close(hip_ctx.normalize_medians(synth["nm_S"], False)[0], synth["nm_false"])
    close(hip_ctx.normalize_medians(synth["nm_pos_S"])[0], synth["nm_pos_auto"])


# ---------------------------------------------------------------- golden: reference fixture
def _pbmc_named(pbmc):
    import plaid_amd
    d, e = pbmc
    X = sp.csc_matrix((d["x"], d["i"], d["p"]), shape=tuple(d["dim"]))
    Xn = plaid_amd.NamedMatrix(X, d["rownames"], d["colnames"])
    return Xn, e


def test_fixture_plaid_via_r_api(pbmc, golden_dir):
    """read.gmt -> gmt2mat -> plaid on the reference's own bundled data (the vignette path,
    vignettes/plaid-vignette.Rmd:49-61), checked against the KAT-pinned oracle outputs."""
    import os
    import plaid_amd
    Xn, e = _pbmc_named(pbmc)
    matG = plaid_amd.gmt2mat(plaid_amd.read_gmt(os.path.join(golden_dir, "hallmarks.gmt")))
    assert matG.shape == (4386, 50)                       # doc/plaid-vignette.html:798
    assert matG.colnames == list(e["G_colnames"])
    S = plaid_amd.plaid(Xn, matG)
    assert S.shape == (50, 50)                            # doc/plaid-vignette.html:809
    close(S.values, e["plaid_norm"])
    close(plaid_amd.plaid(Xn, matG, normalize=False).values, e["plaid_raw"])
    close(plaid_amd.plaid(Xn, matG, stats="sum", normalize=False).values, e["plaid_sum_raw"])
    Xd = plaid_amd.NamedMatrix(Xn.dense(), Xn.rownames, Xn.colnames)
    close(plaid_amd.plaid(Xd, matG).values, e["plaid_dense_norm"])


def test_fixture_ranks_and_replaid(pbmc, golden_dir):
    import os
    import plaid_amd
    Xn, e = _pbmc_named(pbmc)
    matG = plaid_amd.gmt2mat(plaid_amd.read_gmt(os.path.join(golden_dir, "hallmarks.gmt")))
    assert np.array_equal(plaid_amd.sparse_colranks(Xn).values.data, e["sparse_colranks_avg"])
    assert np.array_equal(plaid_amd.sparse_colranks(Xn, ties_method="min").values.data, e["sparse_colranks_min"])
    assert np.array_equal(plaid_amd.colranks(Xn).values, e["colranks_avg"].astype(np.float64))
    assert np.array_equal(plaid_amd.colranks(Xn, ties_method="min").values, e["colranks_min"].astype(np.float64))
    close(plaid_amd.replaid_sing(Xn, matG).values, e["sing"])
    close(plaid_amd.replaid_ssgsea(Xn, matG, alpha=0).values, e["ssgsea_a0"])
    close(plaid_amd.replaid_ssgsea(Xn, matG, alpha=0.25).values, e["ssgsea_a025"])
    Xd = plaid_amd.NamedMatrix(Xn.dense(), Xn.rownames, Xn.colnames)
    close(plaid_amd.replaid_ssgsea(Xd, matG, alpha=0).values, e["ssgsea_dense_a0"])
    close(plaid_amd.replaid_ssgsea(Xd, matG, alpha=0.25).values, e["ssgsea_dense_a025"])


def test_no_overlap_returns_none(hip_ctx, capsys):
    import plaid_amd
    X = plaid_amd.NamedMatrix(np.ones((3, 2)), ["a", "b", "c"], ["s1", "s2"])
    G = plaid_amd.NamedMatrix(sp.csc_matrix(np.ones((2, 1))), ["x", "y"], ["set"])
    assert plaid_amd.plaid(X, G) is None                  # R/plaid.R:66-69
    assert "No overlapping features" in capsys.readouterr().err


# ---------------------------------------------------------------- seeded vs oracle at larger sizes
def test_spmm_seeded_vs_oracle_20k(hip_ctx):
    """20k genes (the LDS-resident limit region), 256 samples, 700 sets, incl. size-500 sets."""
    from plaid_amd import synth as sy
    g, n, m = 20000, 256, 700
    Gp, Gi = sy.geneset_csc(g, m, sort_by_size=False)
    X = sy.dense_columns(g, 0, n)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    exp = _oracle().plaid(X, rn, G, rn)
    close(hip_ctx.plaid_dense(X, Gp, Gi), exp)


@pytest.mark.parametrize("g", [20449, 30001, 45000])
def test_spmm_gene_sliced(hip_ctx, g):
    """g above the LDS-resident limit: the column is consumed in 2-3 gene slices (dense and CSC X)"""
    from plaid_amd import synth as sy
    n, m = 24, 150
    Gp, Gi = sy.geneset_csc(g, m)
    X = sy.dense_columns(g, 0, n)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    exp = _oracle().plaid(X, rn, G, rn)
    close(hip_ctx.plaid_dense(X, Gp, Gi), exp)
    close(hip_ctx.plaid_dense(X, Gp, Gi, "sum", False), _oracle().plaid(X, rn, G, rn, stats="sum", normalize=False))
    Xz = X.copy()
    Xz[np.random.default_rng(1).random(X.shape) < 0.9] = 0.0
    Xs = sp.csc_matrix(Xz)
    close(hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi), _oracle().plaid(Xs, rn, G, rn))


@pytest.mark.parametrize("kernel", ["pair", "single"])
@pytest.mark.parametrize("g,n,m", [(37, 1, 3), (1000, 5, 70), (10224, 9, 130), (10226, 8, 130), (20000, 33, 700),
                                   (25001, 7, 150), (45000, 4, 90), (333, 3, 40)])
def test_spmm_both_dense_kernels(pinned_ctx, kernel, g, n, m):
    """the two-columns-per-pass kernel (1-5 gene slices, odd n) and the one-column kernel give the oracle's scores"""
    from plaid_amd import synth as sy
    hip_ctx = pinned_ctx(spmm_dense_kernel=kernel)
    Gp, Gi = sy.geneset_csc(g, m, kmin=1, kmax=min(g, 400), sort_by_size=False)
    X = sy.dense_columns(g, 0, n) - 8.0
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", False), _oracle().plaid(X, rn, G, rn, normalize=False))
    close(hip_ctx.plaid_dense(X, Gp, Gi, "sum", True), _oracle().plaid(X, rn, G, rn, stats="sum"))
    close(hip_ctx.sing_dense(X, Gp, Gi), _oracle().replaid_sing(X, rn, G, rn))       # alpha/beta epilogue
    Xz = X.copy()
    Xz[np.random.default_rng(g).random(X.shape) < 0.9] = 0.0                           # sparse X, same kernels
    Xs = sp.csc_matrix(Xz)
    close(hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", False),
          _oracle().plaid(Xs, rn, G, rn, normalize=False))


@pytest.mark.parametrize("mode", ["scatter", "gather", "auto"])
@pytest.mark.parametrize("g,n,m,dens", [(500, 3, 40, 0.05), (20000, 9, 700, 0.05), (20000, 5, 24000, 0.03),
                                        (30001, 4, 300, 0.3), (64, 2, 5, 1.0)])
def test_spmm_sparse_x_scatter_and_gather(pinned_ctx, mode, g, n, m, dens):
    """dgCMatrix X: the scatter kernel (work ~ stored values; 1-2 chunks of LDS accumulators), the gather kernel
    and the on-device choice between them all give the oracle's scores, incl. empty columns and the epilogues"""
    from plaid_amd import synth as sy
    hip_ctx = pinned_ctx(spmm_sparse_kernel=mode)
    Gp, Gi = sy.geneset_csc(g, m, kmin=1, kmax=min(g, 300), sort_by_size=False)
    rng = np.random.default_rng(g + m)
    X = np.where(rng.random((g, n)) < dens, np.round(rng.gamma(2.0, 1.0, size=(g, n)), 1), 0.0)
    X[:, n - 1] = 0.0                                           # a sample without stored values
    Xs = sp.csc_matrix(X)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    close(hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", False),
          _oracle().plaid(Xs, rn, G, rn, normalize=False))
    close(hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "sum", True),
          _oracle().plaid(Xs, rn, G, rn, stats="sum"))
    close(hip_ctx.ssgsea_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, 0.25),
          _oracle().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))


@pytest.mark.parametrize("g,n", [(10001, 5), (20001, 3), (333, 2)])
def test_spmm_pair_kernel_odd_genes_strided(g, n):
    """device-level call with ldx = g + 1 (even) and odd g: the last gene of the last slice is staged separately;
    then ldx = g (odd: every other column starts 8 bytes off a 16-byte boundary) through the same kernels"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    m = 90
    rng = np.random.default_rng(g)
    sets = [np.array([0, g // 2, g - 1])]                   # the odd last gene is a member
    sets += [np.sort(rng.choice(g, size=int(k), replace=False)) for k in rng.integers(1, 300, size=m - 1)]
    Gp = np.concatenate([[0], np.cumsum([len(x) for x in sets])]).astype(np.int32)
    Gi = np.concatenate(sets).astype(np.int32)
    X = sy.dense_columns(g, 0, n)
    dev = torch.device("cuda", 0)
    ctx = plaid_amd.Context(0)
    ctx.set_option("spmm_dense_kernel", "pair")
    gs = ctx.geneset(g, Gp, Gi)
    Xd = torch.zeros((n, g + 1), dtype=torch.float64, device=dev)
    Xd[:, :g] = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    Sd = torch.full((n, m + 3), -7.0, dtype=torch.float64, device=dev)
    fl = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.dev_spmm_dense(gs, Xd.data_ptr(), g + 1, n, Sd.data_ptr(), m + 3, "mean", 1.0, 0.0, fl.data_ptr())
    ctx.synchronize()
    S = Sd.cpu().numpy()
    assert np.all(S[:, m:] == -7.0)                          # stride padding untouched
    sizes = np.diff(Gp).astype(float)
    exp = np.stack([np.add.reduceat(X[Gi, j], Gp[:-1]) for j in range(n)]) / (1e-8 + sizes)
    close(S[:, :m], exp)
    # odd leading dimension, and a base pointer that is only 8-byte aligned
    buf = torch.zeros(n * g + 1, dtype=torch.float64, device=dev)
    Xo = buf[1:].view(n, g)
    Xo.copy_(torch.from_numpy(np.ascontiguousarray(X.T)).to(dev))
    Sd.fill_(-7.0)
    torch.cuda.synchronize()
    ctx.dev_spmm_dense(gs, Xo.data_ptr(), g, n, Sd.data_ptr(), m + 3, "mean", 1.0, 0.0, fl.data_ptr())
    ctx.synchronize()
    S2 = Sd.cpu().numpy()
    assert np.array_equal(S2, S)                               # same kernel, same order of additions: same bits
    R = torch.empty((n, g), dtype=torch.float64, device=dev)
    ctx.dev_colranks_dense(Xo.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, 1.0, None)
    Ss = torch.full((n, m), -7.0, dtype=torch.float64, device=dev)
    ctx.dev_spmm_ranks(gs, R.data_ptr(), g, n, Ss.data_ptr(), m, "sum", 1.0, 0.0, fl.data_ptr())
    ctx.synchronize()
    Rh = R.cpu().numpy().T
    close(Ss.cpu().numpy(), np.stack([np.add.reduceat(Rh[Gi, j], Gp[:-1]) for j in range(n)]))
    gs.close()
    ctx.close()


@pytest.mark.parametrize("g", [1, 2, 63, 64, 65, 1000, 4097, 20000, 20448, 20449, 33000])
def test_colranks_sizes_vs_oracle(hip_ctx, g):
    """column lengths around wave/workgroup/LDS boundaries, tied (rounded) data"""
    from oracle import c_oracle
    X = np.round(np.random.default_rng(g).normal(0, 3, size=(g, 5)), 1)
    for tm in ("average", "min"):
        assert np.array_equal(hip_ctx.colranks_dense(X, tm), c_oracle.colranks_dense(X, tm))


def test_colranks_nan_and_negzero(hip_ctx):
    x = np.array([3.0, np.nan, -0.0, 0.0, -2.0, 3.0, np.nan, 1e-300, -1e-300])
    R = hip_ctx.colranks_dense(x.reshape(-1, 1), "average")[:, 0]
    assert np.isnan(R[1]) and np.isnan(R[6])
    assert np.array_equal(R[[0, 2, 3, 4, 5, 7, 8]], np.array([6.5, 3.5, 3.5, 1.0, 6.5, 5.0, 2.0]))


def test_sparse_colranks_ragged(hip_ctx):
    """empty columns, single-entry columns and one long column"""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    lens = [0, 1, 0, 5, 300, 0, 2500, 64, 65, 0]
    Xp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Xx = np.round(rng.gamma(2.0, 1.0, size=Xp[-1]), 1) + 0.1
    for tm in ("average", "min", "max"):
        assert np.array_equal(hip_ctx.colranks_csc(Xp, Xx, tm), c_oracle.sparse_colranks(Xp, Xx, tm))


def test_medians_large_m_select_path(hip_ctx):
    """m above the LDS sort limit takes the radix-select kernel"""
    rng = np.random.default_rng(11)
    S = rng.normal(size=(25000, 6))
    S[rng.random(S.shape) < 0.1] = 0.0
    S = np.abs(S)
    S[:, 3] = 0.0
    exp, _ = _oracle().normalize_medians(S)
    close(hip_ctx.normalize_medians(S)[0], exp)
    exp, _ = _oracle().normalize_medians(S, False)
    close(hip_ctx.normalize_medians(S, False)[0], exp)


def test_streaming_medians_many_columns_per_wavefront(hip_ctx):
    """more columns than the streaming kernel has wavefronts in flight (8 workgroups x 4 per CU): every wavefront
    walks several columns and REUSES its candidate list in global memory -- each column must read back its own keys,
    not what an earlier column of the same wavefront left in a cache; medians differ from column to column"""
    rng = np.random.default_rng(5)
    m, n = 6400, 20480
    S = rng.normal(size=(m, n)) + np.linspace(-3.0, 3.0, n)[None, :]
    S[:, 1::7] = np.round(S[:, 1::7], 1)                         # heavy ties in some columns
    med = np.median(S, axis=0)
    exp = S - med[None, :] + med.mean()
    close(hip_ctx.normalize_medians(S, False)[0], exp)


def test_empty_inputs(hip_ctx):
    Gp = np.zeros(1, dtype=np.int32)
    Gi = np.zeros(0, dtype=np.int32)
    assert hip_ctx.plaid_dense(np.ones((5, 3)), Gp, Gi).shape == (0, 3)          # no sets
    Gp1 = np.array([0, 2], dtype=np.int32)
    Gi1 = np.array([0, 4], dtype=np.int32)
    assert hip_ctx.plaid_dense(np.ones((5, 0)), Gp1, Gi1).shape == (1, 0)         # no samples


def test_bad_arguments_raise(hip_ctx):
    import plaid_amd
    with pytest.raises(plaid_amd.PlaidHipError):
        hip_ctx.plaid_dense(np.ones((5, 3)), np.array([0, 1], dtype=np.int32), np.array([7], dtype=np.int32))
    with pytest.raises(plaid_amd.PlaidHipError):                                   # legal in R, refused: not reproducible
        plaid_amd.colranks(np.ones((4, 2)), ties_method="random")
    with pytest.raises(ValueError):                                                # R: match.arg error
        plaid_amd.colranks(np.ones((4, 2)), ties_method="nonsense")


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("tm", ["first", "last", "dense"])
@pytest.mark.parametrize("g", [1, 7, 300, 4097, 20000, 20352, 25000])
def test_colranks_ties_method_is_passed_through(hip_ctx, tm, g, signed):
    """ties.method goes through to matrixStats::colRanks / base::rank in the reference (R/plaid.R:593,614-617,639-642):
    "first" / "last" / "dense" composed on the device from min-rank passes -- bit-exact against scipy's ordinal / dense
    ranks on the hard columns of the bucket ranker's own test (heavy ties, 95 % zeros, constants, NaN, +-0, infinities)"""
    rng = np.random.default_rng(g)
    X = _rank_cases(g, rng)
    exp = _oracle().colranks(X, signed=signed, ties_method=tm)
    got = hip_ctx.colranks_dense(X, tm, signed)
    ok = ~np.isnan(X)
    for c in range(X.shape[1]):      # NaN: the oracle's rankdata has no NA handling; the device returns NaN there and ranks the rest
        if np.isnan(X[:, c]).any():
            v = X[ok[:, c], c]
            sub = _oracle().colranks(v.reshape(-1, 1), signed=signed, ties_method=tm)[:, 0]
            assert np.isnan(got[~ok[:, c], c]).all() and np.array_equal(got[ok[:, c], c], sub), (c, tm)
        else:
            assert np.array_equal(got[:, c], exp[:, c]), (c, tm)


@pytest.mark.parametrize("tm", ["first", "last"])
def test_sparse_colranks_ties_method_first_and_last(hip_ctx, tm):
    """sparse_colranks -> base::rank(ties.method) on the stored values of every column (R/plaid.R:631-650), ragged columns,
    an empty one, signed and unsigned; "dense" is not a base::rank method and raises like match.arg; colranks() of a sparse
    matrix with its zeros ranked is sparseMatrixStats::colRanks: max / average / min only"""
    import plaid_amd
    rng = np.random.default_rng(3)
    g, n = 5000, 23
    cols, vals = [], []
    for j in range(n):
        k = 0 if j == 5 else int(rng.integers(1, 900))
        cols.append(np.sort(rng.choice(g, k, replace=False)))
        vals.append(np.round(rng.normal(0, 2, k), 0))                    # heavy ties, negatives, stored zeros
    Xp = np.concatenate([[0], np.cumsum([len(c_) for c_ in cols])]).astype(np.int32)
    Xs = sp.csc_matrix((np.concatenate(vals), np.concatenate(cols).astype(np.int32), Xp), shape=(g, n))
    for signed in (False, True):
        got = plaid_amd.sparse_colranks(Xs, signed=signed, ties_method=tm).values
        exp = _oracle().sparse_colranks(Xs, signed=signed, ties_method=tm)
        assert np.array_equal(got.indptr, exp.indptr) and np.array_equal(got.indices, exp.indices)
        assert np.array_equal(got.data, exp.data)
        assert np.array_equal(plaid_amd.colranks(Xs, keep_zero=True, signed=signed, ties_method=tm).values.data, exp.data)
    with pytest.raises(ValueError):
        plaid_amd.sparse_colranks(Xs, ties_method="dense")
    with pytest.raises(ValueError):
        plaid_amd.colranks(Xs, ties_method=tm)                          # sparse = TRUE: sparseMatrixStats takes max / average / min
    # sparse = FALSE on a dgCMatrix: as.matrix(X) through matrixStats, every method legal (R/plaid.R:611-617)
    D = Xs.toarray()
    for t2 in (tm, "dense"):
        assert np.array_equal(plaid_amd.colranks(Xs, sparse=False, ties_method=t2).values, _oracle().colranks(D, ties_method=t2))


# ---------------------------------------------------------------- full-size properties (BASELINE C2)
def test_c2_full_size_properties(hip_ctx):
    """20k genes x 2,048 samples x 5k sets (C2's shape per sample; the sample axis is
    embarrassingly parallel): linearity of the crossprod, column-permutation equivariance,
    and median-normalisation idempotence -- none needs a CPU oracle at this size."""
    from plaid_amd import synth as sy
    g, n, m = 20000, 2048, 5000
    Gp, Gi = sy.geneset_csc(g, m)
    X = sy.dense_columns(g, 0, n)
    S = hip_ctx.plaid_dense(X, Gp, Gi, "mean", False)
    # linearity: plaid(2X + 1) == 2 plaid(X) + k/(k+1e-8)
    k = np.diff(Gp).astype(np.float64)
    S2 = hip_ctx.plaid_dense(2.0 * X + 1.0, Gp, Gi, "mean", False)
    close(S2, 2.0 * S + (k / (k + 1e-8))[:, None])
    # sample permutation equivariance (bit-exact: same per-column arithmetic)
    perm = np.random.default_rng(0).permutation(n)
    Sp = hip_ctx.plaid_dense(np.asfortranarray(X[:, perm]), Gp, Gi, "mean", False)
    assert np.array_equal(Sp, S[:, perm])
    # spot-check 3 columns against the oracle
    from oracle import c_oracle
    close(S[:, [0, 777, 2047]], c_oracle.plaid_dense(X[:, [0, 777, 2047]], Gp, Gi, "mean", False))
    # normalised medians: every column median equals the common value; second pass is a no-op
    N1, _ = hip_ctx.normalize_medians(S)
    med = np.median(N1, axis=0)
    close(med, np.full(n, med.mean()))
    N2, _ = hip_ctx.normalize_medians(N1)
    close(N2, N1)


@pytest.mark.parametrize("m", [1, 2, 3, 64, 65, 1024, 1025, 2048, 2049, 3072, 3073, 4096, 4097, 5000, 5120, 5121, 6144, 6145, 16384, 20000, 33000, 50000, 65536, 65537, 70000])
def test_medians_every_kernel_size_class(hip_ctx, m):
    """column lengths across the register-resident classes (<=2048/6144/16384/32768/65536), the
    radix-select fallback beyond, heavy ties, both parities of the valid count"""
    rng = np.random.default_rng(m)
    n = 5
    S = np.round(rng.normal(size=(m, n)), 2)            # many ties
    S[rng.random(S.shape) < 0.15] = 0.0
    S[:, 1] = np.abs(S[:, 1])
    if m > 3:
        S[:, 2] = 0.0                                    # all masked under ignore.zero
        S[: m // 2, 3] = 7.25                            # the two middle values tie
        S[m // 2:, 3] = -1.5
    for iz in (True, False):
        exp, _ = _oracle().normalize_medians(S, iz)
        got, med = hip_ctx.normalize_medians(S, iz)
        close(got, exp)


@pytest.mark.parametrize("m,n", [(300, 40000), (1500, 20000), (5000, 9000), (6100, 5000)])
def test_medians_wave_kernel_many_columns_per_wavefront(hip_ctx, m, n):
    """the wave-per-column kernel (m <= 6,144) with more columns than wavefronts in the grid: the histogram and the
    trash bins are reused column after column; ties, zeros, NaN columns, both ignore.zero settings; bit-exact medians"""
    from oracle import c_oracle
    rng = np.random.default_rng(m + n)
    S = rng.normal(8.0, 0.2, size=(m, n))
    S[:, ::7] = np.round(S[:, ::7], 1)                   # heavy ties
    S[rng.random(S.shape) < 0.05] = 0.0
    S[:, 5] = 0.0
    S[:, 11] = np.nan
    S[::3, 17] = np.nan
    S[:, 23] = 3.25                                      # constant column: one high dword, exact low-dword range
    S[:, 29] = np.where(rng.random(m) < 0.5, 1.0, np.nextafter(1.0, 2.0))
    for iz in (True, False):
        exp, emed = c_oracle.normalize_medians(S, iz)
        got, med = hip_ctx.normalize_medians(S, iz)
        assert np.array_equal(med, emed, equal_nan=True)
        close(got, exp)


def test_medians_extreme_keys(hip_ctx):
    """low word all ones, +-inf, tiny differences in the last bit"""
    a = np.float64(1.0)
    vals = np.array([a, np.nextafter(a, 2), np.nextafter(a, 0), np.inf, -np.inf, 1e-310, -1e-310,
                     np.frombuffer(np.uint64(0x3FF00000FFFFFFFF).tobytes(), dtype=np.float64)[0],
                     np.frombuffer(np.uint64(0x3FF00000FFFFFFFF).tobytes(), dtype=np.float64)[0], 2.0])
    S = np.stack([vals, vals[::-1], np.sort(vals)], axis=1)
    exp, _ = _oracle().normalize_medians(S, False)
    with np.errstate(all="ignore"):
        close(hip_ctx.normalize_medians(S, False)[0], exp)


def test_device_level_chain_hip_phase_engine():
    """The device-pointer path bench.py and the sharded host use (no host round trips):
    torch tensors -> C ABI dev_* calls on torch's stream, world size 1."""
    import torch
    import plaid_amd
    from plaid_amd import sharded, synth as sy
    g, n, m = 5000, 300, 200
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=300)
    X = sy.dense_columns(g, 0, n, tied=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    eng = sharded.HipPhaseEngine(ctx, gs, dev)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    with torch.cuda.stream(stream):
        S = sharded.sharded_plaid(eng, Xd)
        Ssing = sharded.sharded_sing(eng, Xd)
        Sss = sharded.sharded_ssgsea(eng, Xd, alpha=0.25)
    torch.cuda.synchronize()
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    po = _oracle()
    close(S.cpu().numpy().T, po.plaid(X, rn, G, rn))
    close(Ssing.cpu().numpy().T, po.replaid_sing(X, rn, G, rn))
    close(Sss.cpu().numpy().T, po.replaid_ssgsea(X, rn, G, rn, alpha=0.25))
    gs.close()
    ctx.close()


@pytest.mark.parametrize("g,n,m", [(5000, 301, 200), (7001, 64, 1500)])
def test_sharded_plaid_test_hip_phase_engine(g, n, m):
    """plaid.test in its sample-sharded form (plaid_amd/sharded.py: device-side row sums / sums of squared deviations,
    plaidhip_plaid_test_finish on the host), world size 1, against the one-call entry point plaidhip_plaid_test and the
    oracle; an odd number of genes (the two fold-change columns have an odd leading dimension); a shard split by hand:
    the sums of two half shards added on the host are what two ranks would all-reduce"""
    import torch
    import plaid_amd
    from plaid_amd import sharded, synth as sy
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=300)
    X = sy.dense_columns(g, 0, n, tied=True)
    y = (np.arange(n) % 4 == 1).astype(np.int32)
    X[:300, y == 1] += 0.7
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    eng = sharded.HipPhaseEngine(ctx, gs, dev)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    yd = torch.from_numpy(y).to(dev)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    po = _oracle()
    cols = ["gsetFC", "p.one", "p.two", "p.lm", "p.meta", "q.meta"]
    with torch.cuda.stream(stream):
        for tests, mp_ in ((("one", "two", "lm"), "fisher"), (("one", "lm"), "stouffer"), (("lm",), "fisher")):
            got = sharded.sharded_plaid_test(eng, Xd, yd, Gp, tests, mp_)
            exp = po.plaid_test(X, rn, y, G, rn, None, metap_method=mp_, tests=tests)
            bits = sum({"one": 1, "two": 2, "lm": 4}[t] for t in tests)
            one_call = ctx.plaid_test(X, y, Gp, Gi, None, bits, 0 if mp_ == "fisher" else 1)
            for k, name in enumerate(cols):
                if name in exp:
                    np.testing.assert_allclose(got[:, k], exp[name], rtol=1e-7, atol=1e-300, err_msg=f"{tests} {name}")
                    np.testing.assert_allclose(got[:, k], one_call[:, k], rtol=1e-9, atol=1e-300, err_msg=f"{tests} {name}")
                else:
                    assert np.isnan(got[:, k]).all()
        # two half shards by hand: sums add up; ssd about the global means add up
        h = n // 2
        full = eng.row_group_sums(Xd, yd)
        halves = eng.row_group_sums(Xd[:h].contiguous(), yd[:h].contiguous()) + eng.row_group_sums(Xd[h:].contiguous(), yd[h:].contiguous())
        cnt = torch.tensor([(y == 0).sum(), (y == 1).sum()], dtype=torch.float64, device=dev)
        mean = (full / cnt[:, None]).contiguous()
        ssd_full = eng.row_group_ssd(Xd, yd, mean)
        ssd_halves = eng.row_group_ssd(Xd[:h].contiguous(), yd[:h].contiguous(), mean) + eng.row_group_ssd(Xd[h:].contiguous(), yd[h:].contiguous(), mean)
    torch.cuda.synchronize()
    np.testing.assert_allclose(halves.cpu().numpy(), full.cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(full.cpu().numpy()[1], X[:, y == 1].sum(axis=1), rtol=1e-12)
    np.testing.assert_allclose(ssd_halves.cpu().numpy(), ssd_full.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(ssd_full.cpu().numpy()[0], ((X[:, y == 0] - X[:, y == 0].mean(axis=1, keepdims=True)) ** 2).sum(axis=1), rtol=1e-10)
    gs.close()
    ctx.close()


@pytest.mark.parametrize("g", [300, 7728, 25000])
def test_colranks_csc_dense_result(hip_ctx, g):
    """colranks(sparse X, keep.zero=FALSE): zeros are ranked, dense result (R/plaid.R:602-609);
    g=25000 also exercises the beyond-LDS key scratch together with the densify scratch"""
    from oracle import c_oracle
    rng = np.random.default_rng(g)
    X = np.round(rng.gamma(2.0, 1.0, size=(g, 7)), 1)
    X[rng.random(X.shape) < 0.9] = 0.0
    X[:, 3] = 0.0
    X[::3, 5] *= -1.0
    Xs = sp.csc_matrix(X)
    for tm in ("average", "min"):
        for signed in (False, True):
            got = hip_ctx.colranks_csc_dense(Xs.indptr, Xs.indices, Xs.data, g, tm, signed)
            assert np.array_equal(got, c_oracle.colranks_dense(X, tm, signed))


def _small_named(sparse):
    import plaid_amd
    rng = np.random.default_rng(21)
    g, n, m = 900, 17, 31
    X = np.round(rng.gamma(2.0, 1.5, size=(g, n)), 1)
    X[rng.random(X.shape) < (0.85 if sparse else 0.3)] = 0.0
    rn = [f"g{k}" for k in range(g)]
    grn = [f"g{k}" for k in range(0, g + 60, 1)][30:]          # partial overlap with X's genes
    Gd = (rng.random((len(grn), m)) < 0.04).astype(float)
    Xn = plaid_amd.NamedMatrix(sp.csc_matrix(X) if sparse else X, rn, [f"s{k}" for k in range(n)])
    Gn = plaid_amd.NamedMatrix(sp.csc_matrix(Gd), grn, [f"set{k}" for k in range(m)])
    return Xn, Gn, (sp.csc_matrix(X) if sparse else X), rn, sp.csc_matrix(Gd), grn


@pytest.mark.parametrize("sparse", [False, True])
def test_replaid_ucell_aucell_scse(hip_ctx, sparse):
    """the thin callers of SURVEY.md 8f-1 (R/plaid.R:155-190, 276-309) against the oracle"""
    import plaid_amd
    po = _oracle()
    Xn, Gn, X, rn, G, grn = _small_named(sparse)
    close(plaid_amd.replaid_ucell(Xn, Gn, rmax=200).values, po.replaid_ucell(X, rn, G, grn, rmax=200))
    close(plaid_amd.replaid_aucell(Xn, Gn).values, po.replaid_aucell(X, rn, G, grn))
    close(plaid_amd.replaid_aucell(Xn, Gn, aucMaxRank=120).values, po.replaid_aucell(X, rn, G, grn, auc_max_rank=120))
    for rl in (None, True, False):
        for sm in (False, True):
            close(plaid_amd.replaid_scse(Xn, Gn, removeLog2=rl, scoreMean=sm).values,
                  po.replaid_scse(X, rn, G, grn, remove_log2=rl, score_mean=sm))


# ---------------------------------------------------------------- plaid.test on the device
@pytest.mark.parametrize("metap", ["fisher", "stouffer"])
def test_plaid_test_fixture_matches_oracle_and_vignette(pbmc, golden_dir, metap):
    """plaid.test() (R/plaid.R:392-474) through the R-like API: all three tests and the combined p / q against
    the oracle; with metap.method = "stouffer" the six vignette known answers (doc/plaid-vignette.html:857-869)"""
    import os
    import plaid_amd
    Xn, e = _pbmc_named(pbmc)
    d = pbmc[0]
    matG = plaid_amd.gmt2mat(plaid_amd.read_gmt(os.path.join(golden_dir, "hallmarks.gmt")))
    y = (d["celltype"] == "B").astype(int)
    po = _oracle()
    rn = list(d["rownames"])
    exp = po.plaid_test(Xn.values, rn, y, sp.csc_matrix(matG.values), matG.rownames, None, metap_method=metap,
                        tests=("one", "two", "lm"))
    res = plaid_amd.plaid_test(Xn, y, matG, metap_method=metap, sort_by=None)          # gsetX = NULL: scores stay on the device
    assert res.colnames == ["gsetFC", "p.one", "p.two", "p.lm", "p.meta", "q.meta"] and res.rownames == matG.colnames
    for k, nm in enumerate(res.colnames):
        np.testing.assert_allclose(res.values[:, k], exp[nm], rtol=1e-7, atol=1e-300, err_msg=nm)
    S = plaid_amd.plaid(Xn, matG)
    res2 = plaid_amd.plaid_test(Xn, y, matG, gsetX=S, tests=("one", "lm"), metap_method=metap)   # sorted by p.meta
    assert res2.colnames == ["gsetFC", "p.one", "p.lm", "p.meta", "q.meta"]
    assert np.all(np.diff(res2.values[:, 3]) >= 0)
    if metap == "stouffer":
        # doc/plaid-vignette.html:857-869: head(res) sorted by p.meta, printed with 7 significant digits
        kat = {"HALLMARK_INTERFERON_GAMMA_RESPONSE": (0.003668116, 8.246828e-06, 3.868049e-07),
               "HALLMARK_ALLOGRAFT_REJECTION": (0.102407488, 1.071307e-05, 4.781538e-05),
               "HALLMARK_P53_PATHWAY": (0.038355508, 1.906952e-04, 8.369509e-05),
               "HALLMARK_INTERFERON_ALPHA_RESPONSE": (0.032562973, 9.261621e-03, 1.491854e-03),
               "HALLMARK_PEROXISOME": (0.016625538, 4.052692e-02, 3.080580e-03),
               "HALLMARK_G2M_CHECKPOINT": (0.012385507, 6.049535e-02, 3.638628e-03)}
        assert res2.rownames[:6] == list(kat)
        for k, nm in enumerate(kat):
            np.testing.assert_allclose(res2.values[k, 1:4], kat[nm], rtol=2e-6)
        # html:864-869: q.meta (BH over the 50 sets, R/plaid.R:463), incl. the tied pair
        np.testing.assert_allclose(res2.values[:6, 4], [1.934024e-05, 1.195384e-03, 1.394918e-03, 1.864818e-02,
                                                        3.032190e-02, 3.032190e-02], rtol=2e-6)
        assert res2.values[4, 4] == res2.values[5, 4]


def test_scse_on_the_fixture_takes_the_remove_log2_branch_and_says_so(pbmc, golden_dir, capsys):
    """doc/plaid-vignette.html:920-921 (replaid.scse on the bundled matrix removes the log2): with removeLog2 = NULL the
    decision of R/plaid.R:160-161 is taken ON THE DEVICE (min / max of the stored values and the implicit zeros); it must be
    TRUE on the fixture, the message of :164 must be printed, and the scores must equal the explicit removeLog2 = TRUE run
    and the oracle"""
    import os
    import plaid_amd
    Xn, e = _pbmc_named(pbmc)
    d = pbmc[0]
    matG = plaid_amd.gmt2mat(plaid_amd.read_gmt(os.path.join(golden_dir, "hallmarks.gmt")))
    capsys.readouterr()
    auto = plaid_amd.replaid_scse(Xn, matG)
    msg = capsys.readouterr()
    assert "[replaid.scse] Converting data to linear scale (removing log2)..." in (msg.err + msg.out)
    forced = plaid_amd.replaid_scse(Xn, matG, removeLog2=True)
    assert np.array_equal(auto.values, forced.values)
    capsys.readouterr()
    plain = plaid_amd.replaid_scse(Xn, matG, removeLog2=False)
    msg = capsys.readouterr()
    assert "removing log2" not in (msg.err + msg.out)
    assert np.abs(plain.values - auto.values).max() > 1e-3
    po = _oracle()
    close(auto.values, po.replaid_scse(Xn.values, list(d["rownames"]), sp.csc_matrix(matG.values), matG.rownames))
    # not log-scale data (a value >= 20): the automatic rule leaves X alone, and says nothing
    X2 = Xn.values.copy()
    X2.data = X2.data.copy()
    X2.data[0] = 25.0
    capsys.readouterr()
    a2 = plaid_amd.replaid_scse(plaid_amd.NamedMatrix(X2, Xn.rownames, Xn.colnames), matG)
    msg = capsys.readouterr()
    assert "removing log2" not in (msg.err + msg.out)
    close(a2.values, po.replaid_scse(X2, list(d["rownames"]), sp.csc_matrix(matG.values), matG.rownames, remove_log2=False))


def test_plaid_test_synthetic_and_errors(hip_ctx):
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = 3000, 301, 77
    Gp, Gi = sy.geneset_csc(g, m, kmin=2, kmax=200, sort_by_size=False)
    X = sy.dense_columns(g, 0, n)
    rng = np.random.default_rng(0)
    y = (rng.random(n) < 0.4).astype(int)
    X[:, y == 1] += rng.normal(0, 0.3, size=(g, 1))
    rn = [f"g{k}" for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    cn = [f"s{k}" for k in range(m)]
    exp = _oracle().plaid_test(X, rn, y, G, rn, None, tests=("one", "two", "lm"))
    res = plaid_amd.plaid_test(plaid_amd.NamedMatrix(X, rn, [str(c) for c in range(n)]), y,
                               plaid_amd.NamedMatrix(G, rn, cn), sort_by=None)
    for k, nm in enumerate(res.colnames):
        np.testing.assert_allclose(res.values[:, k], exp[nm], rtol=1e-7, atol=1e-300, err_msg=nm)
    with pytest.raises(ValueError):
        plaid_amd.plaid_test(plaid_amd.NamedMatrix(X, rn, None), y + 1, plaid_amd.NamedMatrix(G, rn, cn))
    with pytest.raises(plaid_amd.PlaidHipError):
        hip_ctx.plaid_test(X, y * 2, Gp, Gi)                     # the C ABI checks y itself (R/plaid.R:394)


@pytest.mark.parametrize("tau", [0, 0.5])
def test_replaid_gsva_matches_oracle(hip_ctx, tau):
    """replaid.gsva (R/plaid.R:338-363): row z-transform, signed ranks, / max|rank|, power, plaid(mean, normalised)"""
    import plaid_amd
    Xn, Gn = _small_named(False)[:2]
    # continuous values: with few distinct values per gene many z-scores of DIFFERENT genes are equal in exact
    # arithmetic (z only depends on the pattern), and which of them ranks first is decided by the last bit of the
    # row mean -- in R as much as here
    X = np.random.default_rng(4).gamma(2.0, 1.5, size=Xn.shape)
    X[:, 3] = X[:, 2]                                               # identical samples are fine
    X[5, :] = 1.25                                                  # a constant gene: sd 0 -> z = 0 (1e-8 guard)
    X[7, :] = X[6, :]                                               # identical genes: exact ties in every sample
    Xn = plaid_amd.NamedMatrix(X, Xn.rownames, Xn.colnames)
    exp = _oracle().replaid_gsva(X, Xn.rownames, sp.csc_matrix(Gn.values), Gn.rownames, tau=tau)
    got = plaid_amd.replaid_gsva(Xn, Gn, tau=tau)
    assert got.rownames == Gn.colnames and got.colnames == Xn.colnames
    close(got.values, exp)
    Xe = np.round(X, 1)                                             # the ECDF variant: ties inside a gene (ties -> max rank)
    Xe[5, :] = 1.25
    Xen = plaid_amd.NamedMatrix(Xe, Xn.rownames, Xn.colnames)
    # (many genes share ECDF values in a sample -> exact ties between genes, which both sides average)
    close(plaid_amd.replaid_gsva(Xen, Gn, tau=tau, rowtf="ecdf").values,
          _oracle().replaid_gsva(Xe, Xn.rownames, sp.csc_matrix(Gn.values), Gn.rownames, tau=tau, rowtf="ecdf"))
    with pytest.raises(ValueError):
        plaid_amd.replaid_gsva(Xn, Gn, rowtf="nope")


def test_spmm_mixed_precision_mode_is_opt_in_and_within_the_bar():
    """plaidhip_set_precision(MIXED): the dense crossprod stages the sample columns as fp32 (sums stay fp64).
    Scores must stay within the 1e-5 relative bar of BASELINE.json (here: elementwise rtol 1e-5 with
    atol 1e-7 for centred scores that cancel towards 0; matrix-relative error ~1e-7), it must really be a
    different path (not bit-identical to fp64), and the default must stay exact fp64."""
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = 20000, 33, 700
    Gp, Gi = sy.geneset_csc(g, m, sort_by_size=False)
    X = sy.dense_columns(g, 0, n)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    po = _oracle()
    ctx = plaid_amd.Context(0)
    exact = ctx.plaid_dense(X, Gp, Gi, "mean", False)
    ctx.set_precision("mixed")
    mixed = ctx.plaid_dense(X, Gp, Gi, "mean", False)
    ref = po.plaid(X, rn, G, rn, normalize=False)
    np.testing.assert_allclose(exact, ref, rtol=1e-12)
    np.testing.assert_allclose(mixed, ref, rtol=1e-5, atol=0)
    rel = np.abs(mixed - ref) / np.abs(ref)
    assert 0 < rel.max() < 2e-7                                   # fp32 input rounding: 2^-24 = 6e-8 per value
    assert np.linalg.norm(mixed - ref) / np.linalg.norm(ref) < 1e-7
    np.testing.assert_allclose(ctx.plaid_dense(X, Gp, Gi, "sum", True), po.plaid(X, rn, G, rn, stats="sum"), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(ctx.sing_dense(X, Gp, Gi), po.replaid_sing(X, rn, G, rn), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ctx.ssgsea_dense(X, Gp, Gi, 0.25), po.replaid_ssgsea(X, rn, G, rn, alpha=0.25), rtol=1e-5, atol=1e-7)
    ctx.set_precision("f64")
    assert np.array_equal(ctx.plaid_dense(X, Gp, Gi, "mean", False), exact)
    ctx.close()


def test_rank_inputs_take_the_fp32_staged_kernel_without_any_rounding(pinned_ctx):
    """replaid.sing / replaid.ssgsea(alpha = 0) / replaid.gsva(tau = 0) feed (half-)integer ranks <= 20,448 to the
    crossprod.  Three stagings exist for them and all must give the fp64 kernels' scores BIT FOR BIT: u16 (2 * rank,
    four samples per LDS entry, integer sums: the default for unsigned ranks), fp32 (exact for such values and for the
    kernel's four-term fp32 partial sums: signed ranks take it), fp64."""
    from plaid_amd import synth as sy
    g, n, m = 20000, 9, 300
    Gp, Gi = sy.geneset_csc(g, m, sort_by_size=False)
    X = sy.dense_columns(g, 0, n, tied=True)
    outs = {}
    for flag in (2, 1, 0):
        hip_ctx = pinned_ctx(ranks_f32=flag)
        outs[flag] = (hip_ctx.sing_dense(X, Gp, Gi), hip_ctx.ssgsea_dense(X, Gp, Gi, 0.0), hip_ctx.gsva(X - 8.0, Gp, Gi, 0.0),
                      hip_ctx.ucell(X, Gp, Gi, np.diff(Gp).astype(float)), hip_ctx.ucell(X, Gp, Gi, np.diff(Gp).astype(float), rmax=99.3))
    for flag in (2, 1):
        for a, b in zip(outs[flag], outs[0]):
            assert np.array_equal(a, b)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    close(outs[2][0], _oracle().replaid_sing(X, rn, G, rn))


@pytest.mark.parametrize("g,n,m", [(8193, 1, 70), (10001, 2, 130), (12345, 3, 64), (20000, 4, 700), (20000, 5, 5000),
                                   (20448, 7, 333), (20447, 13, 65)])
@pytest.mark.parametrize("ties", ["average", "min"])
def test_rank_crossprod_u16_staging_is_bit_identical_to_fp64(g, n, m, ties):
    """plaidhip_dev_spmm_ranks_f64 (u16 staging of 2 * rank, four sample columns per pass, integer sums) against
    plaidhip_dev_spmm_dense_f64 on the same rank matrix: every quad tail (n mod 4), odd and maximal gene counts,
    unsorted gene sets, mean and sum, the affine epilogue of replaid.sing -- equal bit for bit; and against the oracle"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    Gp, Gi = sy.geneset_csc(g, m, kmin=1, kmax=min(400, g), sort_by_size=(m % 2 == 0))
    X = sy.dense_columns(g, 0, n, tied=(ties == "average"))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    ldg = g + (g & 1)
    with torch.cuda.stream(stream):
        Xd = torch.zeros((n, ldg), dtype=torch.float64, device=dev)
        Xd[:, :g] = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
        R = torch.zeros_like(Xd)
        S1 = torch.empty((n, m), dtype=torch.float64, device=dev)
        S2 = torch.empty_like(S1)
        f1 = torch.zeros(4, dtype=torch.int32, device=dev)
        f2 = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.dev_colranks_dense(Xd.data_ptr(), ldg, g, n, R.data_ptr(), ldg, ties, False, 1.0, None)
        for stat, al, be in (("mean", 1.0 / g, -0.5), ("sum", 1.0, 0.0)):
            ctx.dev_spmm_ranks(gs, R.data_ptr(), ldg, n, S1.data_ptr(), m, stat, al, be, f1.data_ptr())
            ctx.dev_spmm_dense(gs, R.data_ptr(), ldg, n, S2.data_ptr(), m, stat, al, be, f2.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(S1, S2), (stat, float((S1 - S2).abs().max()))
            assert torch.equal(f1, f2) and int(f1[3]) == 0
        # not a rank matrix (arbitrary doubles; a value past the u16 range; NaN, Inf and negative "ranks"): the staging sees
        # it and the fp64 launch enqueued behind the speculative one recomputes -- equal to the fp64 route, flags included
        bads = [None, 70000.0, float("nan"), float("inf"), -3.0]
        for bi, bad in enumerate(bads):
            Xb = Xd.clone() if bad is None else R.clone()
            if bad is not None:
                Xb[n - 1, (7 * bi) % g] = bad
                if n > 1:
                    Xb[0, g - 1] = bad
            f1.zero_(); f2.zero_()
            ctx.dev_spmm_ranks(gs, Xb.data_ptr(), ldg, n, S1.data_ptr(), m, "mean", 1.0, 0.0, f1.data_ptr())
            ctx.dev_spmm_dense(gs, Xb.data_ptr(), ldg, n, S2.data_ptr(), m, "mean", 1.0, 0.0, f2.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(torch.nan_to_num(S1, nan=-7.0), torch.nan_to_num(S2, nan=-7.0)), bad
            assert torch.equal(torch.isnan(S1), torch.isnan(S2)) and torch.equal(f1, f2), (bad, f1, f2)
        # and a clean rank matrix afterwards takes the u16 kernel alone again (its private flag words were reset)
        f1.zero_(); f2.zero_()
        ctx.dev_spmm_ranks(gs, R.data_ptr(), ldg, n, S1.data_ptr(), m, "mean", 1.0 / g, -0.5, f1.data_ptr())
        ctx.dev_spmm_dense(gs, R.data_ptr(), ldg, n, S2.data_ptr(), m, "mean", 1.0 / g, -0.5, f2.data_ptr())
        ctx.dev_spmm_dense(gs, R.data_ptr(), ldg, n, S2.data_ptr(), m, "sum", 1.0, 0.0, f2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(f1[:3], torch.tensor([1, 0, 0], dtype=torch.int32, device=dev)) or g < 3   # sing scores < 0 exist
    assert int(f1[3]) == 0
    from oracle import c_oracle
    Ro = c_oracle.colranks_dense(X, ties)
    close(S2.cpu().numpy().T, c_oracle.crossprod_dense(Ro, Gp, Gi, "sum", 4))
    gs.close()
    ctx.close()


# ---------------------------------------------------------------- the bucket ranker against the sorting network
def _rank_cases(g, rng):
    """columns that stress every branch of the bucket ranker: continuous, rounded (heavy ties), 95 % zeros,
    a constant column, two clusters far apart, values two histogram levels cannot separate (fallback to the
    network kernel), +-inf, NaN, negative zero, denormals"""
    cols = [rng.normal(8, 2, g), np.round(rng.normal(8, 2, g), 1),
            np.where(rng.random(g) < 0.95, 0.0, np.round(rng.gamma(2.0, 1.0, g), 1)),
            np.full(g, 3.25), np.where(rng.random(g) < 0.5, 1e-300, 1e300) * rng.random(g),
            np.where(rng.random(g) < 0.5, 1.0, 1.0 + 1e-12 * rng.integers(0, 3, g)),
            rng.choice([-np.inf, np.inf, 0.0, -0.0, 5e-324, -5e-324, 1.0], g),
            np.where(rng.random(g) < 0.01, np.nan, rng.normal(0, 1, g)),
            rng.integers(-3, 4, g).astype(float), np.exp(rng.normal(0, 8, g)),
            1.0 + rng.integers(0, 40, g) * 2.0 ** -52]
    return np.asfortranarray(np.stack(cols, axis=1))


@pytest.mark.parametrize("g", [1, 5, 64, 257, 2048, 2049, 4096, 5000, 8192, 8193, 12288, 12289, 20000, 20352])
def test_bucket_ranker_matches_oracle_on_hard_columns(pinned_ctx, g):
    """every workgroup shape of the bucket ranker, all tie rules, signed and unsigned: bit-exact against the C
    oracle and against the sorting-network kernel (NaN columns: network only, the oracle takes no NaN)"""
    from oracle import c_oracle
    X = _rank_cases(g, np.random.default_rng(g))
    ok = ~np.isnan(X).any(axis=0)
    for tm in ("average", "min", "max"):
        for signed in (False, True):
            got = pinned_ctx(rank_kernel="bucket").colranks_dense(X, tm, signed)
            ref = pinned_ctx(rank_kernel="network").colranks_dense(X, tm, signed)
            assert np.array_equal(got, ref, equal_nan=True), (g, tm, signed)
            if g > 12288:   # the long-column shape before round 4 (512 threads x 40 keys), kept as an option
                alt = pinned_ctx(rank_kernel="bucket512").colranks_dense(X, tm, signed)
                assert np.array_equal(alt, ref, equal_nan=True), (g, tm, signed)
            assert np.array_equal(got[:, ok], c_oracle.colranks_dense(X[:, ok], tm, signed)), (g, tm, signed)
    # sparse ranks (stored values only) through the same kernel
    if g >= 64:
        ctx = pinned_ctx(rank_kernel="bucket")
        lens = np.array([0, 1, g // 3, g, 7, 0, g // 2])
        Xp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        Xx = np.round(np.random.default_rng(g + 1).gamma(2.0, 1.0, size=Xp[-1]), 1) + 0.1
        for tm in ("average", "min"):
            assert np.array_equal(ctx.colranks_csc(Xp, Xx, tm), c_oracle.sparse_colranks(Xp, Xx, tm))


def test_bucket_ranker_power_and_colmax(pinned_ctx):
    """rank^(1 + alpha) fused into the bucket kernel: quarter exponents by square roots, any other by pow()"""
    from plaid_amd import synth as sy
    g, n, m = 20000, 6, 300
    Gp, Gi = sy.geneset_csc(g, m, sort_by_size=False)
    X = sy.dense_columns(g, 0, n, tied=True)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    for alpha in (0.25, 0.5, 1.0, 0.3):
        outs = [pinned_ctx(rank_kernel=k).ssgsea_dense(X, Gp, Gi, alpha) for k in ("bucket", "network", "bucket512")]
        np.testing.assert_allclose(outs[0], outs[1], rtol=1e-12, atol=1e-14)
        assert np.array_equal(outs[0], outs[2])
        close(outs[0], _oracle().replaid_ssgsea(X, rn, G, rn, alpha=alpha))


# ---------------------------------------------------------------- BASELINE target shapes: 20k genes x 50k gene sets
@pytest.fixture(scope="module")
def g50k():
    """the synthetic 50,000-set collection of configs 3-5 (z ~ 6.9e6 memberships), built once per module"""
    from plaid_amd import synth as sy
    g, m = 20000, 50000
    Gp, Gi = sy.geneset_csc(g, m)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    return g, m, Gp, Gi, G, [str(k) for k in range(g)]


@pytest.mark.parametrize("mode", ["scatter", "gather", "auto"])
def test_c3_shape_ssgsea_csc_50k_sets(pinned_ctx, g50k, mode):
    """config 3 per sample: sparse X (5 % stored, ~50 levels per cell), sparse_colranks, alpha = 0.25, 50,000 sets
    = 3 chunks of LDS accumulators in the scatter kernel; every way of multiplying a sparse X, against the oracle"""
    from plaid_amd import synth as sy
    g, m, Gp, Gi, G, rn = g50k
    n = 8
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    ctx = pinned_ctx(spmm_sparse_kernel=mode)
    close(ctx.ssgsea_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, 0.25), _oracle().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))
    close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True), _oracle().plaid(Xs, rn, G, rn))


@pytest.mark.parametrize("sets", [1, 1023, 1024, 1025, 2049, 17407, 17408, 17409, 34816, 34817, 52225])
def test_scatter_kernel_block_interleaved_chunks_at_every_boundary(pinned_ctx, sets):
    """round 6 deals the sets to the LDS chunks in interleaved blocks of 1,024 (block b -> chunk b mod nch, common.h): one set,
    either side of a block, of one / two / three chunks of 17,408 slots, a last block of one set -- with sets in DECREASING
    size (gmt2mat's order, R/gmt-utils.R:25: what made chunks of consecutive sets lopsided) and genes in 1 ... 3 segments of a
    chunk; scatter kernel pinned, mean and sum, raw and normalised, against the oracle"""
    from plaid_amd import synth as sy
    g = 600
    Gp, Gi = sy.geneset_csc(g, sets, kmin=1, kmax=min(g, 90 if sets > 4000 else 400))     # (sorted by decreasing size)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, sets))
    rn = [str(k) for k in range(g)]
    rng = np.random.default_rng(sets)
    Xs = sp.random(g, 24, density=0.3, format="csc", random_state=int(sets) % 1000,
                   data_rvs=lambda k: np.round(rng.gamma(2.0, 1.0, k), 2) + 0.01)
    Xs.sort_indices()
    ctx = pinned_ctx(spmm_sparse_kernel="scatter")
    for stat in ("mean", "sum"):
        close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, stat, False),
              _oracle().plaid(Xs, rn, G, rn, stats=stat, normalize=False))
    close(ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True), _oracle().plaid(Xs, rn, G, rn))


@pytest.mark.parametrize("sets", [700, 24000])
def test_scatter_kernel_any_number_of_stored_values_per_column(pinned_ctx, sets):
    """the scatter kernel's item pipeline (column, chunk, round): columns of 0 ... 6,000 stored values next to each
    other -- empty, one value, one group of 16 +- 1, one round of 1,024 +- 1, two and three rounds +- 1, six rounds --
    with one and two chunks of sets, against the oracle"""
    from plaid_amd import synth as sy
    g = 9000
    Gp, Gi = sy.geneset_csc(g, sets, kmax=300)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, sets))
    rn = [str(k) for k in range(g)]
    counts = [0, 1, 15, 16, 17, 0, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3073, 6000, 0, 5, 1000, 1100, 0]
    rng = np.random.default_rng(11)
    Xp = np.zeros(len(counts) + 1, dtype=np.int32)
    Xi, Xx = [], []
    for j, k in enumerate(counts):
        rows = np.sort(rng.choice(g, size=k, replace=False)).astype(np.int32)
        Xi.append(rows)
        Xx.append(rng.normal(3.0, 2.0, size=k))
        Xp[j + 1] = Xp[j] + k
    Xi = np.concatenate(Xi).astype(np.int32)
    Xx = np.concatenate(Xx)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, len(counts)))
    ctx = pinned_ctx(spmm_sparse_kernel="scatter")
    for stat in ("mean", "sum"):
        close(ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, stat, False), _oracle().plaid(Xs, rn, G, rn, stats=stat, normalize=False))
    close(ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, "mean", True), _oracle().plaid(Xs, rn, G, rn))


def test_scatter_kernel_many_columns_per_workgroup(pinned_ctx):
    """more columns than workgroups: the item pipeline of a workgroup runs across column boundaries (the next
    column's values and segment ranges are requested during the current column's last chunks) -- 1,500 columns of
    0 ... 2,500 stored values, every one against the oracle; also an X without a single stored value"""
    from plaid_amd import synth as sy
    g, sets, n = 3000, 300, 1500
    Gp, Gi = sy.geneset_csc(g, sets, kmax=200)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, sets))
    rn = [str(k) for k in range(g)]
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 2500, size=n)
    counts[::17] = 0
    Xp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    Xi = np.concatenate([np.sort(rng.choice(g, size=int(k), replace=False)) for k in counts]).astype(np.int32)
    Xx = rng.normal(1.0, 2.0, size=int(Xp[-1]))
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    ctx = pinned_ctx(spmm_sparse_kernel="scatter")
    close(ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, "mean", False), _oracle().plaid(Xs, rn, G, rn, normalize=False))
    close(ctx.ssgsea_csc(Xp, Xi, Xx, g, Gp, Gi, 0.25), _oracle().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))
    Xe = sp.csc_matrix((g, 40))
    got = ctx.plaid_csc(Xe.indptr, Xe.indices, Xe.data, g, Gp, Gi, "sum", False)
    assert got.shape == (sets, 40) and not got.any()


def test_scatter_kernel_fixed_point_sums_are_reproducible_and_match_fp64(pinned_ctx):
    """plaidhip_dev_spmm_csc_ranks_f64: rank weights in [0, max(rX)] summed in u64 fixed point by the scatter kernel --
    bit-identical between two runs and between the two item orders (integer sums do not depend on the arrival order of
    the LDS atomics), within 1e-13 relative of the fp64-atomic sums and of the oracle; an input outside [0, rmax] or a NaN
    weight is seen by the sweep on the device and takes the fp64 accumulators (IEEE propagation, as in the reference)"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    g, m, n, alpha = 20000, 24000, 700, 0.25
    Gp, Gi = sy.geneset_csc(g, m)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
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
        for key, fixed, order in (("fx_chunk", "on", "chunk"), ("fx_chunk2", "on", "chunk"), ("fx_col", "on", "column"),
                                  ("f64_chunk", "off", "chunk"), ("f64_col", "off", "column")):
            ctx.set_option("scatter_fixed", fixed)
            ctx.set_option("scatter_order", order)
            ctx.set_option("spmm_sparse_kernel", "scatter")
            S = torch.empty((n, m), dtype=torch.float64, device=dev)
            ctx.dev_spmm_csc_ranks(gs, dp.data_ptr(), di.data_ptr(), Rx.data_ptr(), n, S.data_ptr(), m, gmax.data_ptr(), "mean",
                                   1.0, -0.5, flags.data_ptr(), nnz=len(Xx))
            outs[key] = S
        torch.cuda.synchronize()
        assert int(flags[3]) == 0
        assert torch.equal(outs["fx_chunk"], outs["fx_chunk2"]) and torch.equal(outs["fx_chunk"], outs["fx_col"])
        for key in ("f64_chunk", "f64_col"):
            assert float((outs[key] - outs["fx_chunk"]).abs().max()) < 1e-13
        # a value above rmax / a NaN weight: not fixed-point material -- the fp64 launch of the same call takes it
        ctx.set_option("scatter_fixed", "on")
        for bad in (2.0 * float(gmax[0]), float("nan")):
            Rx2 = Rx.clone()
            Rx2[5] = bad
            S2 = torch.empty((n, m), dtype=torch.float64, device=dev)
            ctx.dev_spmm_csc_ranks(gs, dp.data_ptr(), di.data_ptr(), Rx2.data_ptr(), n, S2.data_ptr(), m, gmax.data_ptr(), "mean",
                                   1.0, -0.5, flags.data_ptr(), nnz=len(Xx))
            ctx.set_option("scatter_fixed", "off")
            S3 = torch.empty((n, m), dtype=torch.float64, device=dev)
            ctx.dev_spmm_csc_ranks(gs, dp.data_ptr(), di.data_ptr(), Rx2.data_ptr(), n, S3.data_ptr(), m, gmax.data_ptr(), "mean",
                                   1.0, -0.5, flags.data_ptr(), nnz=len(Xx))
            ctx.set_option("scatter_fixed", "on")
            torch.cuda.synchronize()
            gene5 = int(Xi[5])
            hit = np.zeros(m, dtype=bool)
            hit[[j for j in range(m) if gene5 in Gi[Gp[j]:Gp[j + 1]]]] = True
            a2, a3 = S2.cpu().numpy(), S3.cpu().numpy()
            col0 = a2[0]          # stored value 5 belongs to sample column 0
            assert Xp[1] > 5
            if bad != bad:
                assert np.isnan(col0[hit]).all() and not np.isnan(col0[~hit]).any() and not np.isnan(a2[1:]).any()
            np.testing.assert_allclose(a2, a3, rtol=0, atol=1e-13, equal_nan=True)
            assert hit.any()
    torch.cuda.synchronize()
    assert int(flags[3]) == 0
    from oracle import fullsize
    _, raw_o = fullsize.ssgsea_csc_raw(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi, alpha, float(gmax[0]))
    np.testing.assert_allclose(outs["fx_chunk"].cpu().numpy().T, raw_o, rtol=0, atol=1e-13)
    gs.close()
    ctx.close()


def test_rank_pipelines_propagate_nan_like_the_fp64_route(hip_ctx, pinned_ctx):
    """NaN in X: matrixStats::colRanks keeps NA and Matrix::crossprod carries it into every set that holds the gene.  The
    default (u16 integer staging of the ranks) must give what the fp64 kernels give: replaid.sing dense and dgCMatrix,
    replaid.ucell, replaid.ssgsea(alpha = 0)"""
    from plaid_amd import synth as sy
    g, n, m = 12000, 9, 300
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=300)
    X = sy.dense_columns(g, 0, n, tied=True)
    X[17, 2] = np.nan
    X[g - 1, 8] = np.nan
    X[5, 0] = np.inf                     # a legal value: ranks last
    ref = pinned_ctx(ranks_f32=0)
    for name, call in (("sing", lambda c: c.sing_dense(X, Gp, Gi)), ("ssgsea0", lambda c: c.ssgsea_dense(X, Gp, Gi, 0.0))):
        a, b = call(hip_ctx), call(ref)
        assert np.isnan(b).any(), name
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        assert np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0)), name
    # which sets: exactly those that hold the gene, in that sample
    S = hip_ctx.sing_dense(X, Gp, Gi)
    has17 = np.array([17 in Gi[Gp[j]:Gp[j + 1]] for j in range(m)])
    assert np.array_equal(np.isnan(S[:, 2]), has17) and not np.isnan(S[:, [0, 1, 3, 4, 5, 6, 7]]).any()
    # dgCMatrix input with a stored NaN
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    Xx = Xx.copy()
    Xx[3] = np.nan
    a = hip_ctx.sing_csc(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi)
    b = ref.sing_csc(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi)
    assert np.isnan(b).any() and np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def test_scatter_kernel_keeps_fp64_sums_for_a_wide_dynamic_range():
    """plaid() on a dgCMatrix whose stored values span many orders of magnitude (raw counts up to 1e6 next to values near
    1; one 1e15 outlier): the fixed-point grid follows the maximum, so small values would lose their relative precision --
    the device-side guard (every score within 2^-40 of the exact sum, or fp64 atomics) must keep such input on the fp64
    accumulators.  Checked against the oracle at 1e-12 relative on every score, which a 2^-4 grid could not meet."""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    from oracle import c_oracle
    g, m, n = 20000, 3000, 64
    Gp, Gi = sy.geneset_csc(g, m)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    ctx.set_option("spmm_sparse_kernel", "scatter")
    rng = np.random.default_rng(5)
    cases = {"counts": np.where(rng.random(len(Xx)) < 0.01, 1e6 * rng.random(len(Xx)), Xx),
             "outlier": Xx.copy(), "tiny": Xx * np.where(rng.random(len(Xx)) < 0.5, 1e-9, 1.0),
             "huge": Xx * 1e301, "zeros": np.where(rng.random(len(Xx)) < 0.3, 0.0, Xx)}
    cases["outlier"][len(Xx) // 2] = 1e15
    with torch.cuda.stream(stream):
        dp, di = (torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (Xp.astype(np.int32), Xi.astype(np.int32)))
        for name, xx in cases.items():
            dx = torch.from_numpy(np.ascontiguousarray(xx)).to(dev)
            S = torch.empty((n, m), dtype=torch.float64, device=dev)
            fl = torch.zeros(4, dtype=torch.int32, device=dev)
            ctx.dev_spmm_csc(gs, dp.data_ptr(), di.data_ptr(), dx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0, fl.data_ptr(),
                             None, nnz=-1)
            torch.cuda.synchronize()
            with np.errstate(all="ignore"):
                exp = c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, xx, g, Gp, Gi, "mean", threads=8)
            got = S.cpu().numpy().T
            ok = exp != 0
            assert np.array_equal(got == 0, exp == 0), name
            assert float(np.max(np.abs(got[ok] / exp[ok] - 1.0))) < 1e-12, name
    gs.close()
    ctx.close()


def test_scatter_kernel_picks_fixed_point_sums_for_nonnegative_values_on_the_device():
    """plaidhip_dev_spmm_csc_f64 (plaid() on a dgCMatrix): nobody declares the values bounded, so a sweep over the stored
    values decides on the device -- all finite and >= 0: u64 fixed-point accumulators (bit-identical between runs and item
    orders, within 1e-13 of the fp64 sums); a negative value, a NaN or an Inf anywhere: fp64 accumulators, with IEEE
    propagation where the oracle has it"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    from oracle import c_oracle
    g, m, n = 20000, 9000, 400
    Gp, Gi = sy.geneset_csc(g, m)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    assert Xx.min() >= 0.0
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    ctx.set_option("spmm_sparse_kernel", "scatter")

    def run(xx, fixed, order):
        ctx.set_option("scatter_fixed", fixed)
        ctx.set_option("scatter_order", order)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        fl = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.dev_spmm_csc(gs, dp.data_ptr(), di.data_ptr(), xx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0, fl.data_ptr(),
                         None, nnz=len(Xx))
        return S, fl

    with torch.cuda.stream(stream):
        dp, di, dx = (torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (Xp.astype(np.int32), Xi.astype(np.int32), Xx))
        a1, f1 = run(dx, "on", "chunk")
        a2, _ = run(dx, "on", "chunk")
        a3, _ = run(dx, "on", "column")
        b1, _ = run(dx, "off", "chunk")
        xneg = dx.clone(); xneg[7] = -1.25
        c1, fneg = run(xneg, "on", "chunk")
        c2, _ = run(xneg, "off", "chunk")
        xnan = dx.clone(); xnan[11] = float("nan"); xnan[13] = float("inf")
        d1, fnan = run(xnan, "on", "chunk")
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(a1, a3)                 # fixed point: no dependence on the arrival order
    assert int(f1[3]) == 0 and int(fneg[3]) == 0 and int(fnan[3]) == 0
    assert float((a1 - b1).abs().max()) < 1e-13
    exp = c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, Xx, g, Gp, Gi, "mean", threads=8)
    np.testing.assert_allclose(a1.cpu().numpy().T, exp, rtol=0, atol=1e-13)
    assert float((c1 - c2).abs().max()) < 1e-12                          # fp64 accumulators either way
    xn = Xx.copy(); xn[7] = -1.25
    np.testing.assert_allclose(c1.cpu().numpy().T, c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, xn, g, Gp, Gi, "mean", threads=8),
                               rtol=0, atol=1e-12)
    xq = Xx.copy(); xq[11] = np.nan; xq[13] = np.inf
    with np.errstate(all="ignore"):
        expq = c_oracle.crossprod_csc(Xp.astype(np.int32), Xi, xq, g, Gp, Gi, "mean", threads=8)
    got = d1.cpu().numpy().T
    assert np.array_equal(np.isnan(got), np.isnan(expq)) and np.array_equal(np.isinf(got), np.isinf(expq))
    ok = np.isfinite(expq)
    np.testing.assert_allclose(got[ok], expq[ok], rtol=0, atol=1e-12)
    gs.close()
    ctx.close()


@pytest.mark.parametrize("rank_kernel", ["bucket", "network"])
def test_c4_shape_ssgsea_and_sing_dense_fp64_50k_sets(pinned_ctx, g50k, rank_kernel):
    """config 4 per sample in the default fp64 mode: dense 20k-gene columns (the register-blocked network / the
    bucket ranker with the fused power), 50,000 sets through the pair kernel, median normalisation at m = 50k"""
    from plaid_amd import synth as sy
    g, m, Gp, Gi, G, rn = g50k
    n = 8
    X = sy.dense_columns(g, 0, n)
    Xt = sy.dense_columns(g, 0, n, tied=True)
    ctx = pinned_ctx(rank_kernel=rank_kernel, ranks_f32=0)        # keep every crossprod on the fp64 kernels
    close(ctx.ssgsea_dense(X, Gp, Gi, 0.25), _oracle().replaid_ssgsea(X, rn, G, rn, alpha=0.25))
    close(ctx.ssgsea_dense(Xt, Gp, Gi, 0.25), _oracle().replaid_ssgsea(Xt, rn, G, rn, alpha=0.25))
    close(ctx.sing_dense(Xt, Gp, Gi), _oracle().replaid_sing(Xt, rn, G, rn))
    close(ctx.plaid_dense(X, Gp, Gi), _oracle().plaid(X, rn, G, rn))


def test_c2_last_columns_of_the_bench_generator(hip_ctx):
    """config 2: plaid() on columns 9,990-9,999 of the very matrix bench.py scores (20k genes x 5k sets)"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    g, m = 20000, 5000
    Gp, Gi = sy.geneset_csc(g, m)
    X = sy.dense_columns(g, 9990, 10000)
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", False), c_oracle.plaid_dense(X, Gp, Gi, "mean", False))
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", True), c_oracle.plaid_dense(X, Gp, Gi, "mean", True))


# ---------------------------------------------------------------- randomised stress (tools/stress_parity.py)
@pytest.mark.parametrize("seed", range(200))
def test_stress_parity_random_case(hip_ctx, seed):
    """one random shape / density / value pattern / precision mode per seed through every host entry point"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "stress_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        failures = mod.run_case(hip_ctx, seed)
    finally:
        hip_ctx.set_precision("f64")
        hip_ctx.set_option("spmm_sparse_kernel", "auto")
    assert not failures, failures


# ---------------------------------------------------------------- stream order: a whole step inside one hipGraph
def test_c3_step_is_capturable_in_a_hip_graph():
    """no plaidhip_dev_* call synchronises or reads anything back (include/plaidhip.h): a full sparse ssGSEA step
    (sparse_colranks -> max -> crossprod -> normalize_medians) is captured in a hipGraph, replayed on new data of
    the same shape, and matches the oracle"""
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = 6000, 40, 900
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=300)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=0.06)
    zx = int(Xp[-1])
    cap = zx + zx // 4                                        # fixed-size buffers: the graph is shape-static
    dXp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    dXi = torch.zeros(cap, dtype=torch.int32, device=dev)
    dXx = torch.zeros(cap, dtype=torch.float64, device=dev)
    dRx = torch.zeros(cap, dtype=torch.float64, device=dev)
    S = torch.zeros((n, m), dtype=torch.float64, device=dev)
    colmax = torch.zeros(n, dtype=torch.float64, device=dev)
    small = torch.zeros(8, dtype=torch.float64, device=dev)      # [0:2] {sum, count}, [2] max(rX)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    med = torch.zeros(n, dtype=torch.float64, device=dev)
    max_nnz = 1024

    def upload(Xp_, Xi_, Xx_):
        dXp.copy_(torch.from_numpy(Xp_.astype(np.int32)))
        dXi[:len(Xi_)].copy_(torch.from_numpy(Xi_.astype(np.int32)))
        dXx[:len(Xx_)].copy_(torch.from_numpy(Xx_))

    def step():
        flags.zero_()
        ctx.dev_colranks_csc(dXp.data_ptr(), dXx.data_ptr(), n, max_nnz, dRx.data_ptr(), "average", False, 1.25,
                             colmax.data_ptr())
        ctx.dev_max(colmax.data_ptr(), n, small.data_ptr() + 16)
        ctx.dev_spmm_csc(gs, dXp.data_ptr(), dXi.data_ptr(), dRx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, -0.5,
                         flags.data_ptr(), small.data_ptr() + 16)            # nnz unknown to the host: decided on the device
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
        ctx.dev_sum(med.data_ptr(), n, small.data_ptr())
        ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, small.data_ptr())

    with torch.cuda.stream(stream):
        upload(Xp, Xi, Xx)
        step()                                                  # warm-up: sizes the context's workspace once
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        step()
    # new data of the same shape class, then replay
    Xp2, Xi2, Xx2 = sy.sparse_columns(g, 100, 100 + n, density=0.06)
    assert int(Xp2[-1]) <= cap and int(np.diff(Xp2).max()) <= max_nnz
    with torch.cuda.stream(stream):
        upload(Xp2, Xi2, Xx2)
    stream.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    Xs2 = sp.csc_matrix((Xx2, Xi2, Xp2), shape=(g, n))
    close(S.cpu().numpy().T, _oracle().replaid_ssgsea(Xs2, rn, G, rn, alpha=0.25))
    gs.close()
    ctx.close()


# ---------------------------------------------------------------- host entry points: pipelined uploads, multi-device form
def test_multi_device_entry_with_one_device_equals_the_context_entry(hip_ctx):
    """plaidhip_*_multi with ndev = 1 (all a 1-GPU box can run) is the same sharded engine as the context entry points:
    bit-identical scores for plaid / sing / ssgsea, dense and dgCMatrix X; and a device list with a repeat is refused"""
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = 9000, 37, 210
    Gp, Gi = sy.geneset_csc(g, m, kmin=3, kmax=300, sort_by_size=False)
    X = sy.dense_columns(g, 0, n, tied=True)
    Xz = np.where(np.random.default_rng(2).random(X.shape) < 0.9, 0.0, X)
    Xs = sp.csc_matrix(Xz)
    assert np.array_equal(plaid_amd.plaid_multi(X, Gp, Gi, "mean", True, devices=1), hip_ctx.plaid_dense(X, Gp, Gi, "mean", True))
    assert np.array_equal(plaid_amd.plaid_multi(X, Gp, Gi, "sum", False, devices=[0]), hip_ctx.plaid_dense(X, Gp, Gi, "sum", False))
    a = plaid_amd.plaid_multi(Xs, Gp, Gi, "mean", True, devices=1)
    close(a, hip_ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True))      # LDS atomics: last bits vary
    assert np.array_equal(plaid_amd.sing_multi(X, Gp, Gi, devices=1), hip_ctx.sing_dense(X, Gp, Gi))
    assert np.array_equal(plaid_amd.ssgsea_multi(X, Gp, Gi, 0.25, devices=1), hip_ctx.ssgsea_dense(X, Gp, Gi, 0.25))
    close(plaid_amd.ssgsea_multi(Xs, Gp, Gi, 0.25, devices=1), hip_ctx.ssgsea_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, 0.25))
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    close(plaid_amd.ssgsea_multi(Xs, Gp, Gi, 0.25, devices=1), _oracle().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))
    with pytest.raises(plaid_amd.PlaidHipError):
        plaid_amd.plaid_multi(X, Gp, Gi, devices=[0, 0])
    plaid_amd.multi_finalize()


@pytest.mark.parametrize("nshards", [2, 3, 5])
def test_multi_device_engine_with_several_shards_on_one_gpu(hip_ctx, nshards):
    """plaidhip_*_multi's engine with ndev >= 2 (what a 1-GPU box cannot reach through the public entry points): a test
    hook runs it with `nshards` contexts on device 0 -- worker threads, rendezvous barriers, the cross-shard max(rX) /
    flags / {sum, count} reductions, shards of unequal size, empty shards (n < nshards), and a shard that fails (the call
    returns an error instead of hanging).  Dense input must equal the one-context result bit for bit; CSC input takes
    the same kernel for every sharding (chosen from the global density) and is compared with the oracle tolerance."""
    import ctypes as C
    import plaid_amd
    from plaid_amd import synth as sy
    from plaid_amd._lib import load
    lib = load()
    fn = lib.plaidhip_debug_sharded_on_one_device
    vp = C.c_void_p
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int, C.c_int,
                   C.c_double, vp]

    def run(method, X, Gp, Gi, stat=0, normalize=1, alpha=0.0, fail=-1, Xcsc=None):
        g = X.shape[0] if Xcsc is None else Xcsc.shape[0]
        n = X.shape[1] if Xcsc is None else Xcsc.shape[1]
        m = len(Gp) - 1
        S = np.full((m, n), np.nan, order="F")
        if Xcsc is None:
            Xf = np.asfortranarray(X)
            rc = fn(0, nshards, fail, method, None, None, Xf.ctypes.data, g, n, Gp.ctypes.data, Gi.ctypes.data, m, stat,
                    normalize, alpha, S.ctypes.data)
        else:
            p_, i_, x_ = (np.ascontiguousarray(Xcsc.indptr, dtype=np.int32), np.ascontiguousarray(Xcsc.indices, dtype=np.int32),
                          np.ascontiguousarray(Xcsc.data, dtype=np.float64))
            rc = fn(0, nshards, fail, method, p_.ctypes.data, i_.ctypes.data, x_.ctypes.data, g, n, Gp.ctypes.data,
                    Gi.ctypes.data, m, stat, normalize, alpha, S.ctypes.data)
        return rc, S

    g, m = 9000, 150
    Gp, Gi = sy.geneset_csc(g, m, kmax=300)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    for n in (1, nshards - 1, 2 * nshards + 1, 37):
        X = sy.dense_columns(g, 0, n, tied=True)
        rc, S = run(0, X, Gp, Gi)
        assert rc == 0 and np.array_equal(S, hip_ctx.plaid_dense(X, Gp, Gi, "mean", True))
        rc, S = run(1, X, Gp, Gi)
        assert rc == 0 and np.array_equal(S, hip_ctx.sing_dense(X, Gp, Gi))
        rc, S = run(2, X, Gp, Gi, alpha=0.25)
        assert rc == 0 and np.array_equal(S, hip_ctx.ssgsea_dense(X, Gp, Gi, 0.25))
        Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
        Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
        rc, S = run(0, None, Gp, Gi, Xcsc=Xs)
        assert rc == 0
        close(S, _oracle().plaid(Xs, rn, G, rn))
        rc, S = run(2, None, Gp, Gi, alpha=0.25, Xcsc=Xs)
        assert rc == 0
        close(S, _oracle().replaid_ssgsea(Xs, rn, G, rn, alpha=0.25))
        rc, S = run(1, None, Gp, Gi, Xcsc=Xs)                          # replaid.sing on CSC: zeros ranked, integer sums
        assert rc == 0 and np.array_equal(S, hip_ctx.sing_dense(Xs.toarray(), Gp, Gi))
    # shards of very different density, a NaN / a negative value in the LAST stored entries of a shard (what decides the
    # accumulator format is swept over exactly the shard's stored values, whatever the whole matrix's density says)
    n = 4 * nshards + 1
    dens = np.where(np.arange(n) % 3 == 0, 0.11, 0.01)
    rng = np.random.default_rng(nshards)
    cols = [np.sort(rng.choice(g, int(g * d), replace=False)).astype(np.int32) for d in dens]
    Xp = np.concatenate([[0], np.cumsum([len(c_) for c_ in cols])]).astype(np.int32)
    Xi = np.concatenate(cols)
    lo_hi = [plaid_amd.shard_bounds(n, nshards, k) for k in range(nshards)]
    for bad in (np.nan, -2.5):
        Xx = np.round(rng.gamma(2.0, 1.0, len(Xi)), 1) + 0.1
        for k in range(nshards):                      # last stored value of every shard's last column
            lo, hi = lo_hi[k]
            if hi > lo:
                Xx[Xp[hi] - 1] = bad
        Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
        rc, S = run(0, None, Gp, Gi, Xcsc=Xs, normalize=0)
        assert rc == 0
        with np.errstate(all="ignore"):
            exp = _oracle().plaid(Xs, rn, G, rn, normalize=False)
        assert np.array_equal(np.isnan(S), np.isnan(exp)), bad
        ok = ~np.isnan(exp)
        np.testing.assert_allclose(S[ok], exp[ok], rtol=1e-10, atol=1e-12)
    # a failing shard: every worker still reaches every rendezvous, the call reports the failure
    X = sy.dense_columns(g, 0, 37)
    for fail in (0, nshards - 1):
        rc, _ = run(2, X, Gp, Gi, alpha=0.25, fail=fail)
        assert rc != 0
        assert b"injected failure" in lib.plaidhip_last_error_string()
    rc, S = run(0, X, Gp, Gi)                                          # and the engine is usable afterwards
    assert rc == 0 and np.array_equal(S, hip_ctx.plaid_dense(X, Gp, Gi, "mean", True))


def test_pipelined_host_upload_many_panels(hip_ctx):
    """a matrix larger than the pinned staging (several 48 MB panels per feeder thread, odd gene count so that the
    device leading dimension differs from nrow): the crossprod per landed panel gives the oracle's scores"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    g, n, m = 19999, 1500, 64                                  # 240 MB of X: ~6 panels
    Gp, Gi = sy.geneset_csc(g, m, sort_by_size=False)
    X = sy.dense_columns(g, 0, n)
    close(hip_ctx.plaid_dense(X, Gp, Gi, "mean", True), c_oracle.plaid_dense(X, Gp, Gi, "mean", True))
    cols = [0, 299, 300, 301, 1499]
    Ssing = hip_ctx.sing_dense(X, Gp, Gi)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    close(Ssing[:, cols], _oracle().replaid_sing(X[:, cols], rn, G, rn))


# ---------------------------------------------------------------- the dense "GEMM" form of config 4 on the matrix cores
def _close_scaled(got, ref, col_scale):
    """the bar SURVEY.md section 7 states for scores that are DIFFERENCES of O(1) numbers (centred by -0.5, shifted by the
    column median): |a - b| <= 1e-5 * max(|b|, col_scale), col_scale = the size of the numbers the difference was made of
    (a scalar or one value per column) -- 1e-5 relative where the score is large, 1e-5 of the column's scale near zero"""
    cs = np.broadcast_to(np.asarray(col_scale, dtype=np.float64), (ref.shape[1],))[None, :]
    err = np.abs(got - ref)
    bound = 1e-5 * np.maximum(np.abs(ref), cs)
    bad = err > bound
    assert not bad.any(), (int(bad.sum()), float(np.nanmax(err / bound)))


@pytest.mark.parametrize("g,n,m", [(20000, 200, 700), (5000, 131, 257), (333, 5, 40)])
def test_mfma_backend_matches_the_spmm_route(pinned_ctx, g, n, m):
    """opt-in alternate backend (PLAIDHIP_OPT_SPMM_DENSE_KERNEL = mfma): dense 0/1 G (bf16, exact) x the bf16 x 3 split
    of the rank weights, fp32 accumulation -- within the 1e-5 bar of the fp64 SpMM route and of the oracle, for plaid(),
    replaid.ssgsea(alpha = 0.25) (the config-4 pipeline) and the sum statistic; tile edges in both dimensions"""
    from plaid_amd import synth as sy
    Gp, Gi = sy.geneset_csc(g, m, kmin=1, kmax=min(g, 400), sort_by_size=False)
    X = sy.dense_columns(g, 0, n)
    rn = [str(k) for k in range(g)]
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    ref = {"plaid": pinned_ctx().plaid_dense(X, Gp, Gi, "mean", False),
           "sum_raw": pinned_ctx().plaid_dense(X, Gp, Gi, "sum", False),
           "sum": pinned_ctx().plaid_dense(X, Gp, Gi, "sum", True),
           "ssgsea": pinned_ctx().ssgsea_dense(X, Gp, Gi, 0.25)}
    ctx = pinned_ctx(spmm_dense_kernel="mfma")
    got = {"plaid": ctx.plaid_dense(X, Gp, Gi, "mean", False), "sum": ctx.plaid_dense(X, Gp, Gi, "sum", True),
           "ssgsea": ctx.ssgsea_dense(X, Gp, Gi, 0.25)}
    np.testing.assert_allclose(got["plaid"], ref["plaid"], rtol=1e-5, atol=0)
    assert 0 < np.max(np.abs(got["plaid"] - ref["plaid"]) / np.abs(ref["plaid"])) < 3e-6      # really the bf16 x 3 / fp32 path
    # normalised sums = (sum over the set) - (column median) + mean of the medians: the scale is the un-normalised sums'
    _close_scaled(got["sum"], ref["sum"], np.max(np.abs(ref["sum_raw"]), axis=0))
    # ssGSEA scores = mean(rank weight) / max - 0.5, median-shifted: differences of numbers of size 0.5
    _close_scaled(got["ssgsea"], ref["ssgsea"], 0.5)
    _close_scaled(got["ssgsea"], _oracle().replaid_ssgsea(X, rn, G, rn, alpha=0.25), 0.5)


def test_mfma_backend_at_config_4_width_50k_sets(pinned_ctx, g50k):
    """config 4 is named "MFMA path": the dense-G backend at the full 50,000 sets (2 GB of bf16 G) x 20,000 genes on 2,048
    samples -- replaid.ssgsea(alpha = 0.25), ranks and medians included, and the sum statistic -- against the SpMM route
    and the oracle, at |a - b| <= 1e-5 * max(|b|, column scale)"""
    from plaid_amd import synth as sy
    g, m, Gp, Gi, G, rn = g50k
    n = 2048
    X = sy.dense_columns(g, 0, n)
    ref = pinned_ctx().ssgsea_dense(X, Gp, Gi, 0.25)
    ref_sum = pinned_ctx().plaid_dense(X, Gp, Gi, "sum", True)
    ref_sum_raw_scale = np.max(np.abs(pinned_ctx().plaid_dense(X[:, :64], Gp, Gi, "sum", False)), axis=0).min()
    ctx = pinned_ctx(spmm_dense_kernel="mfma")
    got = ctx.ssgsea_dense(X, Gp, Gi, 0.25)
    got_sum = ctx.plaid_dense(X, Gp, Gi, "sum", True)
    _close_scaled(got, ref, 0.5)
    assert 0 < float(np.max(np.abs(got - ref))) < 1e-6                  # really the bf16 x 3 / fp32 path, inside its bound
    _close_scaled(got_sum, ref_sum, ref_sum_raw_scale)                  # (the smallest of 64 columns' largest sums: ~4e3)
    assert float(np.max(np.abs(got_sum - ref_sum))) > 0
    cols = np.r_[0:24, n - 24:n]                                        # oracle on a sample of the columns: max(rX) is
    exp = _oracle().replaid_ssgsea(X[:, cols], rn, G, rn, alpha=0.25)   # g^1.25 in every tie-free column, the mean of the
    raw = got[:, cols] - got[:, cols].mean(axis=0, keepdims=True)       # medians is not: compare up to the column shift
    _close_scaled(raw, exp - exp.mean(axis=0, keepdims=True), 0.5)
