"""The score matrix's way home through the host entry points (plaid_amd/csrc/multi.cpp, HomeBuffer): results of more than
16 MB return in 64 MB chunks behind threads that make the caller's pages.  The bytes must not depend on it: fresh and
reused destinations, destinations at odd byte offsets, sizes just around the chunk boundaries, every host entry.
`pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(g, n, m, density=0.05):
    from plaid_amd import synth as sy
    Gp, Gi = sy.geneset_csc(g, m, kmax=60)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=density)
    return Gp, Gi, Xp.astype(np.int32), Xi, Xx


@pytest.mark.parametrize("m,n", [(2048, 1024), (2048, 1025), (8192, 1024), (8191, 1025), (4100, 4100), (9000, 5000)])
def test_plaid_csc_result_is_the_same_through_any_destination(hip_ctx, m, n):
    """16 MB exactly (the plain copy's last size), one byte range more, 64 MB exactly, odd sizes, 134 MB (three chunks),
    360 MB; the destination fresh from the allocator, reused, and starting 8 bytes / 4,088 bytes into a page"""
    g = 3000
    Gp, Gi, Xp, Xi, Xx = _inputs(g, n, m)
    ref = hip_ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    again = hip_ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, out=ref.copy(order="F"))
    assert np.array_equal(ref, again)
    for off in (1, 511):
        buf = np.full(m * n + 1024, 7.0)
        out = buf[off:off + m * n].reshape((m, n), order="F")
        got = hip_ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, out=out)
        assert got is out and np.array_equal(ref, got)
        assert (buf[:off] == 7.0).all() and (buf[off + m * n:] == 7.0).all()          # nothing outside the result is touched
    # against the device-pointer path on a few columns (the oracle comparison of this entry lives in test_gpu_parity.py)
    from oracle import c_oracle
    raw = hip_ctx.plaid_csc(Xp, Xi, Xx, g, Gp, Gi, "mean", False)
    cols = [0, n // 2, n - 1]
    sub_p = np.zeros(len(cols) + 1, np.int32)
    sub_i, sub_x = [], []
    for k, c in enumerate(cols):
        sub_i.append(Xi[Xp[c]:Xp[c + 1]]); sub_x.append(Xx[Xp[c]:Xp[c + 1]])
        sub_p[k + 1] = sub_p[k] + Xp[c + 1] - Xp[c]
    exp = c_oracle.crossprod_csc(sub_p, np.concatenate(sub_i), np.concatenate(sub_x), g, Gp, Gi, "mean", threads=4)
    np.testing.assert_allclose(raw[:, cols], exp, rtol=1e-12, atol=1e-14)


def test_dense_entries_and_rank_matrices_come_home_whole(hip_ctx):
    """plaid_dense / colranks (a g x n rank matrix of 80 MB) / replaid.sing through the same way home, against the oracle"""
    from oracle import c_oracle
    from plaid_amd import synth as sy
    g, n, m = 5000, 2100, 3000
    Gp, Gi = sy.geneset_csc(g, m, kmax=80)
    X = sy.dense_columns(g, 0, n, tied=True)
    R = hip_ctx.colranks_dense(X, ties="average")
    assert R.shape == (g, n)
    assert np.array_equal(R, c_oracle.colranks_dense_mt(X, "average", threads=4))
    S = hip_ctx.plaid_dense(X, Gp, Gi, "mean", False)
    cols = [0, 1, n - 1]
    exp = c_oracle.crossprod_dense(np.asfortranarray(X[:, cols]), Gp, Gi, "mean", threads=4)
    np.testing.assert_allclose(S[:, cols], exp, rtol=1e-12, atol=1e-14)
    out = np.full((m, n), np.nan, order="F")
    S2 = hip_ctx.plaid_dense(X, Gp, Gi, "mean", False, out=out)
    assert S2 is out and np.array_equal(S, S2)
