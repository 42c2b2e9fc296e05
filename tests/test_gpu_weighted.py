"""chunked_crossprod() (R/plaid.R:100-123) with a GENERAL x: `t(x) %*% y` for a sparse x whose stored values differ inside a
column (weighted / signed sets).  The reference function is generic (`Matrix::crossprod(x, y)`); plaid() only ever gives it
the column-scaled 0/1 matrix, which is the scheduled-kernel path tested in test_gpu_parity.py.  Oracle:
oracle/plaid_oracle.chunked_crossprod (scipy's sparse product, the same sums in sequential order) -- `pytest -m gpu`.
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


def _weights(rng, g, m, dens=0.02):
    """signed, non-uniform weights; one empty column, one single-entry column, one dense column, explicit zeros"""
    W = sp.random(g, m, density=dens, format="csc", random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  data_rvs=lambda k: np.round(rng.normal(0.0, 2.0, size=k), 3))
    W = W.tolil()
    if m > 3:
        W[:, 1] = 0.0
        W[:, 2] = 0.0
        W[g // 2, 2] = -1.5
        W[:, 3] = rng.normal(size=(g, 1))
    W = W.tocsc()
    W.sort_indices()
    if W.nnz > 5:
        W.data[::17] = 0.0        # stored zeros stay stored entries
    return W


@pytest.mark.parametrize("g,n,m", [(700, 9, 77), (20480, 5, 33), (20481, 7, 33), (33000, 3, 12), (1200, 1100, 5), (64, 1, 1)])
@pytest.mark.parametrize("ysparse", [False, True])
def test_weighted_crossprod_vs_oracle(hip_ctx, g, n, m, ysparse):
    """LDS-resident (g <= 20,480) and global-gather (above) columns, more columns than workgroups (n = 1,100), dense and
    dgCMatrix y"""
    rng = np.random.default_rng(g * 7 + n + m)
    W = _weights(rng, g, m)
    Y = np.round(rng.gamma(2.0, 1.5, size=(g, n)), 2) - 1.0
    if ysparse:
        Y[rng.random(Y.shape) < 0.9] = 0.0
        if n > 2:
            Y[:, 1] = 0.0           # an empty column
        Ys = sp.csc_matrix(Y)
        got = hip_ctx.crossprod_weighted(W.indptr, W.indices, W.data, g, Yp=Ys.indptr, Yi=Ys.indices, Yx=Ys.data)
        exp = _oracle().chunked_crossprod(W, Ys)
    else:
        got = hip_ctx.crossprod_weighted(W.indptr, W.indices, W.data, g, Y=Y)
        exp = _oracle().chunked_crossprod(W, Y)
    assert got.shape == (m, n)
    close(got, exp)
    if m > 3:
        assert np.all(got[1, :] == 0.0)                     # empty column of x: exactly 0


def test_weighted_crossprod_nan_only_reaches_the_sets_that_hold_the_gene(hip_ctx):
    """structural zeros of x are skipped (Matrix::crossprod of a dgCMatrix): a NaN in y touches the columns of x that
    store that row -- stored zeros included -- and no other"""
    rng = np.random.default_rng(5)
    g, n, m = 500, 6, 40
    W = _weights(rng, g, m, dens=0.05)
    Y = rng.normal(size=(g, n))
    Y[123, 2] = np.nan
    got = hip_ctx.crossprod_weighted(W.indptr, W.indices, W.data, g, Y=Y)
    holds = np.asarray([123 in W.indices[W.indptr[j]:W.indptr[j + 1]] for j in range(m)])
    assert holds.any() and not holds.all()
    assert np.all(np.isnan(got[holds, 2])) and not np.any(np.isnan(got[~holds, 2]))
    assert not np.any(np.isnan(np.delete(got, 2, axis=1)))
    Y[123, 2] = 0.25
    close(np.delete(got, 2, axis=1), np.delete(_oracle().chunked_crossprod(W, Y), 2, axis=1))


def test_weighted_crossprod_device_pointers_and_leading_dimensions(hip_ctx):
    """the device-level entry with ldy > g and lds > m (a panel of larger matrices); rows of S behind m stay untouched"""
    import torch
    rng = np.random.default_rng(11)
    g, n, m, ldy, lds = 3000, 40, 55, 3008, 64
    W = _weights(rng, g, m)
    Y = rng.normal(size=(g, n))
    dev = torch.device("cuda", 0)
    Yd = torch.zeros((n, ldy), dtype=torch.float64, device=dev)
    Yd[:, :g] = torch.from_numpy(np.ascontiguousarray(Y.T)).to(dev)
    Sd = torch.full((n, lds), -7.0, dtype=torch.float64, device=dev)
    Wp = torch.from_numpy(W.indptr.astype(np.int32)).to(dev)
    Wi = torch.from_numpy(W.indices.astype(np.int32)).to(dev)
    Wx = torch.from_numpy(W.data.astype(np.float64)).to(dev)
    torch.cuda.synchronize()
    hip_ctx.dev_crossprod_weighted(Wp.data_ptr(), Wi.data_ptr(), Wx.data_ptr(), g, m, Yd.data_ptr(), ldy, n,
                                   Sd.data_ptr(), lds)
    hip_ctx.synchronize()
    S = Sd.cpu().numpy()
    close(S[:, :m].T, _oracle().chunked_crossprod(W, Y))
    assert np.all(S[:, m:] == -7.0)
    # y as CSC through the device-level entry
    Ys = sp.csc_matrix(np.where(rng.random(Y.shape) < 0.8, 0.0, Y))
    Yp = torch.from_numpy(Ys.indptr.astype(np.int32)).to(dev)
    Yi = torch.from_numpy(Ys.indices.astype(np.int32)).to(dev)
    Yx = torch.from_numpy(Ys.data.astype(np.float64)).to(dev)
    Sd.fill_(-7.0)
    torch.cuda.synchronize()
    hip_ctx.dev_crossprod_weighted_csc(Wp.data_ptr(), Wi.data_ptr(), Wx.data_ptr(), g, m, Yp.data_ptr(), Yi.data_ptr(),
                                       Yx.data_ptr(), n, Sd.data_ptr(), lds)
    hip_ctx.synchronize()
    S = Sd.cpu().numpy()
    close(S[:, :m].T, _oracle().chunked_crossprod(W, Ys))
    assert np.all(S[:, m:] == -7.0)


@pytest.mark.parametrize("ysparse", [False, True])
def test_chunked_crossprod_r_api_membership_and_weighted(hip_ctx, ysparse, capsys):
    """the R-like entry: a binary x and a column-scaled binary x take the membership kernels, a weighted x the general
    kernel; the chunk loop and its message are the reference's (R/plaid.R:107-119)"""
    import plaid_amd
    po = _oracle()
    rng = np.random.default_rng(3)
    g, n, m = 800, 23, 19
    Y = np.round(rng.gamma(2.0, 1.5, size=(g, n)), 1)
    if ysparse:
        Y[rng.random(Y.shape) < 0.85] = 0.0
    rn = [f"g{k}" for k in range(g)]
    Yn = plaid_amd.NamedMatrix(sp.csc_matrix(Y) if ysparse else Y, rn, [f"s{k}" for k in range(n)])
    B = sp.csc_matrix((rng.random((g, m)) < 0.05).astype(float))
    scaled = B @ sp.diags(1.0 / (1e-8 + np.asarray(B.sum(axis=0)).ravel()))      # what plaid() builds (:74-77)
    Wt = _weights(rng, g, m, dens=0.05)
    for x in (B, sp.csc_matrix(scaled), Wt):
        xn = plaid_amd.NamedMatrix(x, rn, [f"set{k}" for k in range(m)])
        exp = po.chunked_crossprod(x, sp.csc_matrix(Y) if ysparse else Y)
        got = plaid_amd.chunked_crossprod(xn, Yn)
        assert got.rownames == xn.colnames and got.colnames == Yn.colnames
        close(got.values, exp)
        capsys.readouterr()
        close(plaid_amd.chunked_crossprod(xn, Yn, chunk=7).values, exp)           # 4 chunks, the last one short
        assert "[chunked_crossprod] chunked compute: chunk = 7" in capsys.readouterr().err
    with pytest.raises(ValueError):
        plaid_amd.chunked_crossprod(plaid_amd.NamedMatrix(B[:-1, :], rn[:-1], [f"set{k}" for k in range(m)]), Yn)


def test_weighted_crossprod_empty_and_bad_arguments(hip_ctx):
    import plaid_amd
    W = sp.csc_matrix((40, 0))
    assert hip_ctx.crossprod_weighted(W.indptr, W.indices, W.data, 40, Y=np.ones((40, 3))).shape == (0, 3)
    W = sp.csc_matrix(np.eye(40)[:, :5])
    assert hip_ctx.crossprod_weighted(W.indptr, W.indices, W.data, 40, Y=np.ones((40, 0))).shape == (5, 0)
    bad = W.indices.copy()
    bad[2] = 40                                                    # row index outside [0, g)
    with pytest.raises(plaid_amd.PlaidHipError):
        hip_ctx.crossprod_weighted(W.indptr, bad, W.data, 40, Y=np.ones((40, 2)))
