"""Medians selected inside the DENSE crossprod launch (plaidhip_dev_spmm_dense_fused_f64 + plaidhip_dev_col_medians_resume,
round 5): normalize_medians (R/plaid.R:554-575) without a second pass over the score matrix, for the fp64 pair kernel.  The
workgroup of a column pair computes the pair's mean scores from the X it stages and the tile ends of the last gene slice
classify the scores they write.  The contract is exactness: scores and flags equal the plain crossprod's bit for bit, every
median equals what the standalone kernels select on the same S -- whatever the bracket prediction does; the tests also look
at HOW MANY columns the fused path resolved, so that a silent all-fallback is seen.  `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    ctx.set_option("fused_medians", "on")       # (by default only from 1e9 scores on: these matrices are smaller)
    yield torch, dev, stream, ctx
    ctx.close()


def _run(ctx, torch, dev, stream, gs, m, X, ldx, n, alpha=1.0, beta=0.0, alpha_div=None, ignore_zero=None, stat="mean"):
    """(S fused, S plain, flags fused, flags plain, med fused, med plain, status, cal, pred) on the same dense input"""
    with torch.cuda.stream(stream):
        S1 = torch.empty((n, m), dtype=torch.float64, device=dev)
        S2 = torch.empty((n, m), dtype=torch.float64, device=dev)
        f1 = torch.zeros(4, dtype=torch.int32, device=dev)
        f2 = torch.zeros(4, dtype=torch.int32, device=dev)
        med1 = torch.full((n,), 12345.0, dtype=torch.float64, device=dev)
        med1s = torch.full((n,), 777.0, dtype=torch.float64, device=dev)
        med2 = torch.full((n,), 54321.0, dtype=torch.float64, device=dev)
        div = alpha_div.data_ptr() if alpha_div is not None else None
        token = ctx.dev_spmm_dense_fused(gs, X.data_ptr(), ldx, n, S1.data_ptr(), m, stat, alpha, beta, f1.data_ptr(), div)
        ctx.dev_col_medians_resume(S1.data_ptr(), m, m, n, ignore_zero, med1.data_ptr(), f1.data_ptr(), token=token)
        ctx.dev_col_medians(S1.data_ptr(), m, m, n, ignore_zero, med1s.data_ptr(), f1.data_ptr())
        ctx.dev_spmm_dense(gs, X.data_ptr(), ldx, n, S2.data_ptr(), m, stat, alpha, beta, f2.data_ptr(), div)
        ctx.dev_col_medians(S2.data_ptr(), m, m, n, ignore_zero, med2.data_ptr(), f2.data_ptr())
    torch.cuda.synchronize()
    nf, p_status, p_cal, pending = ctx.dev_fused_medians_info()
    status = np.zeros(max(nf, 1), dtype=np.int32)
    cal = np.zeros(4)
    if nf:
        ctx.lib.plaidhip_memcpy_d2h(ctx.handle, status.ctypes.data_as(C.c_void_p), C.c_void_p(p_status), C.c_size_t(4 * nf))
        ctx.lib.plaidhip_memcpy_d2h(ctx.handle, cal.ctypes.data_as(C.c_void_p), C.c_void_p(p_cal), C.c_size_t(32))
    assert not pending
    assert np.array_equal(med1.cpu().numpy(), med1s.cpu().numpy(), equal_nan=True), "fused medians differ from the standalone kernels on the same S"
    return S1, S2, f1.cpu().numpy(), f2.cpu().numpy(), med1.cpu().numpy(), med2.cpu().numpy(), status[:nf], cal, token


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _normal(torch, dev, n, g, seed, ld=None):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    ld = ld or g
    X = torch.zeros((n, ld), dtype=torch.float64, device=dev)
    X[:, :g] = torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen) * 2.0 + 8.0
    return X


@pytest.mark.parametrize("g,m", [(20000, 7000), (20000, 24000), (20000, 50000), (9000, 24000), (15001, 9000), (30000, 9000)])
def test_dense_fused_medians_are_bit_identical_and_mostly_resolved(env, g, m):
    """plaid() on dense N(8, 2^2) columns: one, two (an odd slice too) and three gene slices; scores, flags and medians equal
    the plain route's bit for bit; the bracket around (mean from the staged X + calibrated offset) resolves nearly every column"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    n = 1300
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    with torch.cuda.stream(stream):
        X = _normal(torch, dev, n, g, 5)
    S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, X, g, n)
    assert token > 0 and torch.equal(S1, S2) and np.array_equal(f1, f2)
    assert _same(m1, m2)
    assert len(status) == n and status.mean() > 0.9, (status.mean(), cal)
    assert cal[1] > 0 and cal[2] == 0.0
    gs.close()


def test_dense_fused_medians_ssgsea_weights_sum_statistic_and_a_leading_dimension(env):
    """config 4's pipeline (colranks ^ 1.25, alpha / max(rX), beta = -0.5) with ldx > g; then stats = "sum" on the raw matrix"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n, ld = 20000, 24000, 1300, 20006
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    with torch.cuda.stream(stream):
        X = torch.round(_normal(torch, dev, n, g, 7, ld) * 10.0) / 10.0         # ~170-fold ties in every column
        R = torch.zeros_like(X)
        colmax = torch.zeros(n, dtype=torch.float64, device=dev)
        gm = torch.zeros(1, dtype=torch.float64, device=dev)
        ctx.dev_colranks_dense(X.data_ptr(), ld, g, n, R.data_ptr(), ld, "average", False, 1.25, colmax.data_ptr())
        ctx.dev_max(colmax.data_ptr(), n, gm.data_ptr())
    S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, R, ld, n, 1.0, -0.5, gm)
    assert token > 0 and torch.equal(S1, S2) and np.array_equal(f1, f2) and _same(m1, m2)
    assert status.mean() > 0.9 and f1[0] == 1
    S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, X, ld, n, stat="sum")
    assert token > 0 and torch.equal(S1, S2) and _same(m1, m2)
    # (sums of 15 ... 500 values: the spread of a column's scores is far wider than the calibration sample's deviations, the
    # bracket catches more than its share and overflows where it does: what matters is exactness, checked in _run)
    gs.close()


def test_dense_fused_medians_tie_heavy_scores_zero_scores_and_nan_columns(env):
    """non-negative integer data with whole rows of zeros: many exactly equal scores and exact zeros, none negative ->
    min(x) == 0 -> the zeros are masked (R/plaid.R:556-563); explicit ignore.zero either way; a few columns hold a NaN
    (NaN scores are skipped, na.rm): those go to the standalone kernel, the medians stay exact"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 12010, 30000, 1300
    Gp, Gi = sy.geneset_csc_real(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    with torch.cuda.stream(stream):
        X = torch.floor(torch.rand((n, g), dtype=torch.float64, device=dev, generator=gen) * 4.0)      # 0 ... 3
        X *= (torch.rand((n, g), dtype=torch.float64, device=dev, generator=gen) < 0.06)                # 94 % zeros
    for iz in (None, True, False):
        S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, X, g, n, ignore_zero=iz)
        assert token > 0 and torch.equal(S1, S2) and np.array_equal(f1, f2) and f1[1] == 1 and f1[0] == 0
        assert _same(m1, m2), iz
        assert cal[2] == 1.0
        if iz is False:
            assert status.sum() == 0            # the other rule than calibrated: nothing is taken from the candidates
    with torch.cuda.stream(stream):
        Xn = _normal(torch, dev, n, g, 9)
        Xn[5, 17] = float("nan")
        Xn[700, 3] = float("nan")
        Xn[1299, g - 1] = float("inf")
    S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, Xn, g, n)
    assert token > 0 and torch.equal(torch.isnan(S1), torch.isnan(S2)) and np.array_equal(f1, f2) and f1[2] == 1
    assert torch.equal(torch.nan_to_num(S1, nan=0.0, posinf=1.0, neginf=-1.0), torch.nan_to_num(S2, nan=0.0, posinf=1.0, neginf=-1.0))
    assert _same(m1, m2)
    assert status[5] == 0 and status[700] == 0 and status.mean() > 0.8
    gs.close()


def test_dense_fused_entry_is_the_plain_route_when_it_does_not_apply(env):
    """few sets per column, few columns, the one-column kernel, rank inputs that take a compact staging, the size rule:
    plaidhip_dev_spmm_dense_fused_f64 + ..._resume are exactly the plain pair"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g = 20000
    for m, n, kernel, fm in ((3000, 1300, "auto", "on"), (24000, 300, "auto", "on"), (24000, 1300, "single", "on"),
                             (24000, 1300, "auto", "auto"), (24000, 1300, "auto", "off")):
        ctx.set_option("spmm_dense_kernel", kernel)
        ctx.set_option("fused_medians", fm)
        Gp, Gi = sy.geneset_csc(g, m)
        gs = ctx.geneset(g, Gp, Gi)
        with torch.cuda.stream(stream):
            X = _normal(torch, dev, n, g, 21)
        S1, S2, f1, f2, m1, m2, status, cal, token = _run(ctx, torch, dev, stream, gs, m, X, g, n)
        assert token == 0 and len(status) == 0
        assert torch.equal(S1, S2) and np.array_equal(f1, f2) and _same(m1, m2)
        gs.close()
    ctx.set_option("spmm_dense_kernel", "auto")
    ctx.set_option("fused_medians", "on")


def test_dense_fused_predicted_means_are_the_column_means(env):
    """the workgroup's own prediction -- alpha * sum_i x[i, c] u[i] + beta * kappa from the staged X -- IS the mean of the
    column's scores (to rounding): what the bracket is centred on"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 20000, 9000, 1301                     # (an odd number of columns: the last pair has one)
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    with torch.cuda.stream(stream):
        X = _normal(torch, dev, n, g, 31)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        fl = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        ref = torch.empty(n, dtype=torch.float64, device=dev)
        token = ctx.dev_spmm_dense_fused(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 0.5, -0.25, fl.data_ptr(), None)
        assert token > 0
        # the scratch starts with pred[n]: its address is info[2] (cal) - up(8 n) ... read it through the status pointer's base
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), fl.data_ptr(), token=token)
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, ref.data_ptr(), fl.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(med, ref)
    nf, p_status, p_cal, _ = ctx.dev_fused_medians_info()
    assert nf == n
    pred = np.zeros(n)
    p_pred = p_cal - ((8 * n + 255) // 256) * 256
    ctx.lib.plaidhip_memcpy_d2h(ctx.handle, pred.ctypes.data_as(C.c_void_p), C.c_void_p(p_pred), C.c_size_t(8 * n))
    np.testing.assert_allclose(pred, S.mean(dim=1).cpu().numpy(), rtol=1e-11, atol=1e-13)
    gs.close()


def test_dense_fused_entry_points_check_their_arguments(env):
    """bad dimensions / null pointers / a negative token are error codes with a message (no launch, no crash): the same
    contract as every other entry point"""
    import plaid_amd
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 3000, 7000, 8
    Gp, Gi = sy.geneset_csc(g, m, kmin=5, kmax=60)
    gs = ctx.geneset(g, Gp, Gi)
    with torch.cuda.stream(stream):
        X = _normal(torch, dev, n, g, 1)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        fl = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
    for bad in (dict(ldx=g - 1), dict(lds=m - 1), dict(n=-1), dict(X=None), dict(S=None)):
        a = dict(X=X.data_ptr(), ldx=g, n=n, S=S.data_ptr(), lds=m)
        a.update(bad)
        with pytest.raises(plaid_amd.PlaidHipError):
            ctx.dev_spmm_dense_fused(gs, a["X"], a["ldx"], a["n"], a["S"], a["lds"], "mean", 1.0, 0.0, fl.data_ptr(), None)
    with pytest.raises(plaid_amd.PlaidHipError):
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), fl.data_ptr(), token=-5)
    with pytest.raises(plaid_amd.PlaidHipError):                      # ignore.zero = auto needs the flag words
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), None, token=0)
    # and a legal small call still works afterwards (ineligible by size: the plain route, token 0)
    with torch.cuda.stream(stream):
        token = ctx.dev_spmm_dense_fused(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, fl.data_ptr(), None)
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), fl.data_ptr(), token=token)
    torch.cuda.synchronize()
    assert token == 0 and bool(torch.isfinite(med).all())
    gs.close()
