"""Dense ranks of sparse columns WITHOUT densifying them (colranks(X sparse, keep.zero = FALSE), R/plaid.R:602-609: zeros
are ranked, dense result) and replaid.sing on a dgCMatrix (R/plaid.R:213-219).  All zeros of a column tie, so its dense
ranks follow from the ranks of the stored values; the oracle ranks the densified matrix.  `pytest -m gpu`."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-9


def _sparse_matrix(rng, g, n, dens):
    X = np.round(rng.normal(0.5, 2.0, size=(g, n)), 1)          # negatives, heavy ties
    X[rng.random(X.shape) >= dens] = 0.0
    if n > 3:
        X[:, 1] = 0.0                                            # an empty column
        X[:, 2] = np.abs(X[:, 2])                                # no negatives
        X[: g // 3, 3] = -1.5                                    # many tied negatives
    Xs = sp.csc_matrix(X)
    if Xs.nnz > 10:
        Xs.data[::23] = 0.0                                      # stored zeros stay stored
    return Xs


@pytest.mark.parametrize("g,n,dens", [(300, 9, 0.2), (7728, 7, 0.1), (20352, 5, 0.05), (25000, 6, 0.08), (36601, 5, 0.06),
                                      (70000, 3, 0.03), (50, 4, 1.0)])
def test_colranks_csc_dense_any_number_of_rows(hip_ctx, g, n, dens):
    """the host entry picks the stored-values route (every column here stores <= 20,352 values): ranks must equal the dense
    ranks of the densified matrix bit for bit, for every ties method, signed or not, beyond the 20,352-row limit of the
    rank kernel too"""
    from oracle import c_oracle
    rng = np.random.default_rng(g + n)
    Xs = _sparse_matrix(rng, g, n, dens)
    X = Xs.toarray()
    for tm in ("average", "min", "max"):
        for signed in (False, True):
            got = hip_ctx.colranks_csc_dense(Xs.indptr, Xs.indices, Xs.data, g, tm, signed)
            assert np.array_equal(got, c_oracle.colranks_dense(X, tm, signed)), (tm, signed)


def test_colranks_csc_dense_nz_device_entry_power_colmax_and_nan(hip_ctx):
    """device-level entry: ldr > g, fused power (by square roots and by pow), column maxima, NaN among the stored values,
    and the refusal of a column with more stored values than one pass ranks"""
    import torch
    import plaid_amd
    from oracle import c_oracle
    rng = np.random.default_rng(8)
    g, n, ldr = 27001, 6, 27004
    Xs = _sparse_matrix(rng, g, n, 0.07)
    X = Xs.toarray()
    dev = torch.device("cuda", 0)
    Xp = torch.from_numpy(Xs.indptr.astype(np.int32)).to(dev)
    Xi = torch.from_numpy(Xs.indices.astype(np.int32)).to(dev)
    Xx = torch.from_numpy(Xs.data.astype(np.float64)).to(dev)
    Rx = torch.empty(Xs.nnz, dtype=torch.float64, device=dev)
    max_nnz = int(np.diff(Xs.indptr).max())
    for power in (1.0, 1.25, 1.3):
        for tm in ("average", "min"):
            R = torch.full((n, ldr), -7.0, dtype=torch.float64, device=dev)
            cm = torch.empty(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            hip_ctx.dev_colranks_csc_dense_nz(Xp.data_ptr(), Xi.data_ptr(), Xx.data_ptr(), g, n, max_nnz, Rx.data_ptr(),
                                              R.data_ptr(), ldr, tm, False, power, cm.data_ptr())
            hip_ctx.synchronize()
            Rh = R.cpu().numpy()
            assert np.all(Rh[:, g:] == -7.0)
            with np.errstate(invalid="ignore"):
                exp = c_oracle.colranks_dense(X, tm, False) ** power
            np.testing.assert_allclose(Rh[:, :g].T, exp, rtol=1e-13, atol=0)
            np.testing.assert_allclose(cm.cpu().numpy(), np.nanmax(exp, axis=0), rtol=1e-13)
    # NaN among the stored values: set aside (NaN rank), the others ranked among themselves -- as the dense kernel does
    # (pinned by test_colranks_nan_and_negzero); the plain-C oracle takes NaN-free vectors only
    Xs.data[5] = np.nan
    Xs.data[Xs.indptr[4] + 1] = np.nan
    for tm in ("average", "min", "max"):
        for signed in (False, True):
            got = hip_ctx.colranks_csc_dense(Xs.indptr, Xs.indices, Xs.data, g, tm, signed)
            assert np.array_equal(got, hip_ctx.colranks_dense(Xs.toarray(), tm, signed), equal_nan=True), (tm, signed)
    assert np.isnan(got).sum() == 2
    with pytest.raises(plaid_amd.PlaidHipError):
        hip_ctx.dev_colranks_csc_dense_nz(Xp.data_ptr(), Xi.data_ptr(), Xx.data_ptr(), g, n, 20353, Rx.data_ptr(), R.data_ptr(),
                                          ldr, "min", False, 1.0, None)


def test_colranks_csc_dense_long_columns_still_take_the_densify_route(hip_ctx):
    """a column with more than 20,352 stored values: the host entry densifies and ranks it as before"""
    from oracle import c_oracle
    rng = np.random.default_rng(3)
    g, n = 24000, 3
    X = np.round(rng.normal(size=(g, n)), 2)
    X[rng.random(X.shape) < 0.05] = 0.0
    Xs = sp.csc_matrix(X)
    assert np.diff(Xs.indptr).max() > 20352
    assert np.array_equal(hip_ctx.colranks_csc_dense(Xs.indptr, Xs.indices, Xs.data, g, "min", False),
                          c_oracle.colranks_dense(X, "min", False))


@pytest.mark.parametrize("g,n,m", [(900, 17, 31), (33538, 40, 120), (20000, 300, 60)])
def test_replaid_sing_on_a_sparse_matrix(hip_ctx, g, n, m):
    """replaid.sing with a dgCMatrix X through the R-like API: the CSC slots go to the device (plaidhip_sing_csc); scores
    equal those of the dense route and of the oracle (which densifies, as the reference does)"""
    import plaid_amd
    from oracle import plaid_oracle as po
    rng = np.random.default_rng(g)
    X = np.round(rng.gamma(2.0, 1.5, size=(g, n)), 1)
    X[rng.random(X.shape) < 0.92] = 0.0
    rn = [f"g{k}" for k in range(g)]
    Gd = sp.random(g, m, density=min(0.5, 60.0 / g), format="csc", random_state=np.random.RandomState(5))
    Gd.data[:] = 1.0
    Xn = plaid_amd.NamedMatrix(sp.csc_matrix(X), rn, [f"s{k}" for k in range(n)])
    Xd = plaid_amd.NamedMatrix(X, rn, Xn.colnames)
    Gn = plaid_amd.NamedMatrix(Gd, rn, [f"set{k}" for k in range(m)])
    got = plaid_amd.replaid_sing(Xn, Gn)
    assert got.rownames == Gn.colnames and got.colnames == Xn.colnames
    np.testing.assert_allclose(got.values, po.replaid_sing(X, rn, Gd, rn), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(got.values, plaid_amd.replaid_sing(Xd, Gn).values, rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------- dense columns beyond the LDS: value-partitioned ranking
def _long_columns(rng, g, n, kind):
    if kind == "normal":
        X = rng.normal(8.0, 2.0, size=(g, n))
    elif kind == "ties":
        X = np.round(rng.normal(0.0, 3.0, size=(g, n)), 1)                  # ~200 distinct values
    elif kind == "zeros":                                                     # a dense single-cell matrix: 94 % zeros
        X = np.where(rng.random((g, n)) < 0.06, np.round(rng.gamma(2.0, 1.5, size=(g, n)), 1), 0.0)
    elif kind == "cluster":                                                   # 70 % of the keys inside one narrow interval
        X = rng.normal(0.0, 1.0, size=(g, n))                                 # that no sample quantile isolates: fallback
        sel = rng.random((g, n)) < 0.7
        X[sel] = 5.0 + 1e-9 * rng.random(int(sel.sum()))
    elif kind == "constant":
        X = np.full((g, n), 2.5)
    elif kind == "adversarial":                                               # the 1,024 sampled rows hold the small values:
        X = 100.0 + rng.random((g, n))                                        # every splitter lies below the bulk, whose open
        rows = (np.arange(1024, dtype=np.int64) * g) >> 10                    # interval is far too long -> device-side fallback
        X[rows, :] = rng.random((1024, n))
        X[:, n - 1] = rng.normal(size=g)                                      # (and one ordinary column next to them)
    else:
        raise ValueError(kind)
    return X


@pytest.mark.parametrize("g,kind", [(20353, "normal"), (25000, "ties"), (33538, "zeros"), (36601, "normal"), (60000, "ties"),
                                    (61000, "cluster"), (150000, "normal"), (40000, "constant"), (33000, "zeros"),
                                    (45000, "adversarial")])
def test_colranks_dense_long_columns_by_value_partition(hip_ctx, g, kind):
    """columns longer than the bucket ranker's 20,352 keys are cut by value into segments it takes: ranks equal the
    oracle's bit for bit -- every ties method, signed, with a heavy single value (zeros), with clustered values that send
    the column to the sorting-network fallback, and for a constant column"""
    from oracle import c_oracle
    rng = np.random.default_rng(g)
    n = 5 if g < 100000 else 2
    X = _long_columns(rng, g, n, kind)
    if kind != "constant":
        X[::7, 0] *= -1.0
    for tm in ("average", "min", "max"):
        for signed in (False, True):
            assert np.array_equal(hip_ctx.colranks_dense(X, tm, signed), c_oracle.colranks_dense(X, tm, signed)), (tm, signed)


def test_colranks_dense_long_columns_power_colmax_nan_and_strides(hip_ctx):
    """device-level entry on 30,001-row columns: leading dimensions larger than the column, fused power, column maxima,
    NaN entries (set aside), and more columns than workgroups in flight"""
    import torch
    from oracle import c_oracle
    rng = np.random.default_rng(12)
    g, n, ldx, ldr = 30001, 700, 30004, 30002
    X = np.round(rng.normal(1.0, 2.0, size=(g, n)), 2)
    X[rng.random(X.shape) < 0.3] = 0.0
    dev = torch.device("cuda", 0)
    Xd = torch.zeros((n, ldx), dtype=torch.float64, device=dev)
    Xd[:, :g] = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    for power, tm, signed in ((1.0, "min", False), (1.25, "average", False), (1.3, "max", False), (1.25, "average", True)):
        R = torch.full((n, ldr), -7.0, dtype=torch.float64, device=dev)
        cm = torch.empty(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        hip_ctx.dev_colranks_dense(Xd.data_ptr(), ldx, g, n, R.data_ptr(), ldr, tm, signed, power, cm.data_ptr())
        hip_ctx.synchronize()
        Rh = R.cpu().numpy()
        assert np.all(Rh[:, g:] == -7.0)
        base = c_oracle.colranks_dense(X, tm, signed)
        exp = np.sign(base) * np.abs(base) ** power
        np.testing.assert_allclose(Rh[:, :g].T, exp, rtol=1e-13, atol=0)
        np.testing.assert_allclose(cm.cpu().numpy(), np.abs(exp).max(axis=0), rtol=1e-13)
    Xn = X[:, :4].copy()
    Xn[17, 0] = np.nan
    Xn[::1000, 2] = np.nan
    got = hip_ctx.colranks_dense(Xn, "average", False)
    assert np.array_equal(np.isnan(got), np.isnan(Xn))
    for c in range(4):
        ok = ~np.isnan(Xn[:, c])
        assert np.array_equal(got[ok, c], c_oracle.colranks_dense(Xn[ok, c:c + 1], "average", False)[:, 0])


@pytest.mark.parametrize("signed", [False, True])
def test_sparse_colranks_columns_with_more_stored_values_than_the_lds_holds(hip_ctx, signed):
    """sparse_colranks() (R/plaid.R:631-650) on a dgCMatrix whose columns store 25,000-45,000 values (a dense-ish matrix kept
    in sparse form) next to a short and an empty column: the value-partitioned route on CSC columns, bit-exact"""
    from oracle import c_oracle
    rng = np.random.default_rng(31)
    g = 60000
    lens = [45000, 0, 300, 25000, 20353, 33333]
    cols, vals = [], []
    for k in lens:
        cols.append(np.sort(rng.choice(g, size=k, replace=False)))
        v = np.round(rng.normal(0.0, 3.0, size=k), 1)
        v[v == 0.0] = 0.5
        vals.append(v)
    Xp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Xx = np.concatenate(vals)
    for tm in ("average", "min", "max"):
        got = hip_ctx.colranks_csc(Xp, Xx, tm, signed)
        assert np.array_equal(got, c_oracle.sparse_colranks(Xp, Xx, tm, signed)), tm


def test_colranks_dense_long_columns_several_scratch_panels(pinned_ctx):
    """more columns than one 2-GiB scratch panel of the value-partitioned route holds (60,000 rows: 1,789 columns per
    panel): bit-identical to the sorting-network route (rank_kernel = network), which the shorter tests pin to the oracle"""
    import torch
    g, n = 60000, 2100
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    X = torch.round(torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen) * 50) / 10      # ties
    X[:, ::3] = 0.0
    R1 = torch.empty_like(X)
    R2 = torch.empty_like(X)
    c1 = torch.empty(n, dtype=torch.float64, device=dev)
    c2 = torch.empty(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx = pinned_ctx()
    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R1.data_ptr(), g, "average", False, 1.25, c1.data_ptr())
    ctx.set_option("rank_kernel", "network")
    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R2.data_ptr(), g, "average", False, 1.25, c2.data_ptr())
    ctx.synchronize()
    # (the network kernel raises to the power with pow(), the bucket ranker by square roots: a few ulp)
    assert torch.allclose(R1, R2, rtol=1e-13, atol=0) and torch.allclose(c1, c2, rtol=1e-13, atol=0)
    ctx.set_option("rank_kernel", "auto")
    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R1.data_ptr(), g, "min", True, 1.0, None)
    ctx.set_option("rank_kernel", "network")
    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R2.data_ptr(), g, "min", True, 1.0, None)
    ctx.synchronize()
    assert torch.equal(R1, R2)


def test_sharded_sing_csc_hip_phase_engine():
    """the torch.distributed form (plaid_amd/sharded.py) of replaid.sing on a CSC shard, single process: panels of dense
    min-ranks from the stored values + the exact rank crossprod; equals the host entry's scores bit for bit"""
    import torch
    import plaid_amd
    from plaid_amd import sharded, synth as sy
    from oracle import plaid_oracle as po
    g, n, m = 24001, 37, 90
    rng = np.random.default_rng(2)
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=300)
    X = np.round(rng.gamma(2.0, 1.5, size=(g, n)), 1)
    X[rng.random(X.shape) < 0.9] = 0.0
    Xs = sp.csc_matrix(X)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    gs = ctx.geneset(g, Gp, Gi)
    eng = sharded.HipPhaseEngine(ctx, gs, dev)
    shard = sharded.CscShard.from_scipy(Xs, 0, n, device=dev)
    with torch.cuda.stream(stream):
        S = sharded.sharded_sing_csc(eng, shard, panel_bytes=10 * (g + 1) * 8)      # 4 panels of 10 cells
    torch.cuda.synchronize()
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    np.testing.assert_allclose(S.cpu().numpy().T, po.replaid_sing(X, rn, G, rn), rtol=RTOL, atol=ATOL)
    assert np.array_equal(S.cpu().numpy().T, ctx.sing_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi))
    gs.close()
    ctx.close()


def test_colranks_dense_long_columns_extreme_values(pinned_ctx):
    """+-inf, +-0, denormals, the largest and smallest doubles, values one ulp apart, an all-NaN column and a column with a
    single non-NaN entry in 26,000-row columns: the value-partitioned route equals the sorting-network route bit for bit
    (NaN positions included), and the oracle on the NaN-free columns"""
    from oracle import c_oracle
    rng = np.random.default_rng(77)
    g, n = 26000, 6
    X = rng.normal(size=(g, n))
    special = np.array([np.inf, -np.inf, 0.0, -0.0, 5e-324, -5e-324, 1.7976931348623157e308, -1.7976931348623157e308,
                        1.0, np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0), 2.2250738585072014e-308])
    X[rng.choice(g, size=4000, replace=False), 0] = rng.choice(special, size=4000)
    X[:, 1] = rng.choice(special, size=g)                        # only 12 distinct values (+-0 tie)
    X[:, 2] = np.nan
    X[:, 3] = np.nan
    X[777, 3] = -3.5
    X[::5, 4] = np.nan
    ctx = pinned_ctx()
    for tm in ("average", "min", "max"):
        for signed in (False, True):
            ctx.set_option("rank_kernel", "auto")
            with np.errstate(all="ignore"):
                a = ctx.colranks_dense(X, tm, signed)
                ctx.set_option("rank_kernel", "network")
                b = ctx.colranks_dense(X, tm, signed)
            assert np.array_equal(a, b, equal_nan=True), (tm, signed)
            assert np.array_equal(np.isnan(a), np.isnan(X))
            assert np.array_equal(a[:, [0, 1, 5]], c_oracle.colranks_dense(X[:, [0, 1, 5]], tm, signed)), (tm, signed)
