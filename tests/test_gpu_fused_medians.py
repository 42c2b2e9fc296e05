"""Medians selected inside the sparse crossprod launch (plaidhip_dev_spmm_csc_fused_f64 + plaidhip_dev_col_medians_resume,
round 4): normalize_medians (R/plaid.R:554-575) without a second pass over the score matrix.  The contract is exactness:
every median equals what the standalone kernels select -- bit for bit -- whatever the bracket prediction does; these tests
also look at HOW MANY columns the fused path resolved, so that a silent all-fallback is seen.  `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, alpha=1.0, beta=0.0, ranks=False, ignore_zero=None, stat="mean"):
    """(S raw, flags, med_fused, med_plain, status) -- crossprod + medians through the fused pair of entries and through the
    plain pair, on the same inputs"""
    with torch.cuda.stream(stream):
        dp, di, dx = (torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (Xp.astype(np.int32), Xi.astype(np.int32), Xx))
        S1 = torch.empty((n, m), dtype=torch.float64, device=dev)
        S2 = torch.empty((n, m), dtype=torch.float64, device=dev)
        f1 = torch.zeros(4, dtype=torch.int32, device=dev)
        f2 = torch.zeros(4, dtype=torch.int32, device=dev)
        med1 = torch.full((n,), 12345.0, dtype=torch.float64, device=dev)
        med2 = torch.full((n,), 54321.0, dtype=torch.float64, device=dev)
        rmax = None
        vals = dx
        if ranks:   # replaid.ssgsea's weights: sparse_colranks ^ 1.25, max(rX) on the device
            vals = torch.empty_like(dx)
            colmax = torch.zeros(n, dtype=torch.float64, device=dev)
            gm = torch.zeros(1, dtype=torch.float64, device=dev)
            ctx.dev_colranks_csc(dp.data_ptr(), dx.data_ptr(), n, int(np.diff(Xp).max()), vals.data_ptr(), "average", False, 1.25,
                                 colmax.data_ptr())
            ctx.dev_max(colmax.data_ptr(), n, gm.data_ptr())
            rmax = gm.data_ptr()
        ctx.dev_spmm_csc_fused(gs, dp.data_ptr(), di.data_ptr(), vals.data_ptr(), n, S1.data_ptr(), m, stat, alpha, beta,
                               f1.data_ptr(), None, rmax, nnz=len(Xx))
        ctx.dev_col_medians_resume(S1.data_ptr(), m, m, n, ignore_zero, med1.data_ptr(), f1.data_ptr())
        # the standalone kernels on the very matrix the fused launch wrote: what med1 has to equal bit for bit, whatever the
        # arrival order of fp64 atomics did to the last bits of S (the fixed-point sums do not depend on it)
        med1s = torch.full((n,), 777.0, dtype=torch.float64, device=dev)
        ctx.dev_col_medians(S1.data_ptr(), m, m, n, ignore_zero, med1s.data_ptr(), f1.data_ptr())
        if ranks:
            ctx.dev_spmm_csc_ranks(gs, dp.data_ptr(), di.data_ptr(), vals.data_ptr(), n, S2.data_ptr(), m, rmax, stat, alpha, beta,
                                   f2.data_ptr(), nnz=len(Xx))
        else:
            ctx.dev_spmm_csc(gs, dp.data_ptr(), di.data_ptr(), vals.data_ptr(), n, S2.data_ptr(), m, stat, alpha, beta,
                             f2.data_ptr(), None, nnz=len(Xx))
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
    return S1, S2, f1.cpu().numpy(), f2.cpu().numpy(), med1.cpu().numpy(), med2.cpu().numpy(), status[:nf], cal


@pytest.fixture(scope="module")
def env():
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    ctx.set_option("spmm_sparse_kernel", "scatter")
    ctx.set_option("fused_medians", "on")       # (by default only from 1e9 scores on: these matrices are smaller)
    yield torch, dev, stream, ctx
    ctx.close()


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("m", [7000, 24000, 50000])
def test_fused_medians_ssgsea_weights_are_bit_identical_and_mostly_resolved(env, m):
    """config 3's pipeline (rank weights, alpha / max(rX), beta = -0.5): scores, flags and medians equal the plain route's
    bit for bit; the bracket around (predicted column mean + calibrated offset) resolves nearly every column"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, n = 20000, 1500
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, 1.0, -0.5, ranks=True)
    assert torch.equal(S1, S2) and np.array_equal(f1, f2)
    assert _same(m1, m2)
    assert len(status) == n and status.mean() > 0.9, (status.mean(), cal)
    assert cal[1] > 0 and cal[2] == 0.0
    gs.close()


def test_fused_medians_plaid_counts_with_zero_scores_ignore_zero_rule(env):
    """plaid() on non-negative sparse data: exact zero scores, no negative ones -> min(x) == 0 -> the zeros are masked
    (R/plaid.R:556-563).  The calibration columns tell the rule, the medians of the non-zero scores are selected among the
    candidates; explicit ignore.zero = FALSE / TRUE on the same matrix: the first contradicts the calibration -> every column
    falls back, still exact"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 12010, 30000, 1200
    Gp, Gi = sy.geneset_csc_real(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n, density=0.06)
    for iz in (None, True, False):
        S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, ignore_zero=iz)
        assert torch.equal(S1, S2) and np.array_equal(f1, f2) and f1[1] == 1 and f1[0] == 0
        assert _same(m1, m2), iz
        assert cal[2] == 1.0
        if iz is False:
            assert status.sum() == 0                    # the other rule than calibrated: nothing is taken from the candidates
        else:
            assert status.mean() > 0.8, (iz, status.mean(), cal)
    gs.close()


def test_fused_medians_hard_columns(env):
    """columns the bracket cannot serve -- empty cells (all scores equal), cells with a NaN (NaN scores are skipped, na.rm),
    cells 100x denser or with a few huge values (mean far from the median), duplicated cells (ties everywhere) -- fall back
    or resolve, the medians stay those of the standalone kernel; calibration columns included among the odd ones"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 20000, 20000, 1100
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    rng = np.random.default_rng(11)
    cols, vals = [], []
    for j in range(n):
        if j % 97 == 5:
            k = 0                                                         # empty cell
        elif j % 97 == 9:
            k = 8000                                                      # very dense cell
        else:
            k = int(rng.integers(600, 1400))
        rows = np.sort(rng.choice(g, k, replace=False)).astype(np.int32)
        v = np.round(rng.gamma(2.0, 1.0, k), 1) + 0.1
        if j % 97 == 13 and k:
            v[rng.integers(0, k, 3)] = 1e6                                # outliers: the mean leaves the median behind
        if j % 97 == 21 and k:
            v[0] = np.nan
        if j % 97 == 30 and j > 0:
            rows, v = cols[-1], vals[-1]                                  # a duplicated cell
        cols.append(rows)
        vals.append(v)
    Xp = np.concatenate([[0], np.cumsum([len(c_) for c_ in cols])]).astype(np.int32)
    Xi, Xx = np.concatenate(cols), np.concatenate(vals)
    for iz in (None, False):
        S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n, ignore_zero=iz)
        # (a NaN among the stored values: fp64 atomics, sums in arrival order -- two launches agree to ~1e-16 relative, and so
        # do their medians; the exactness of the selection is checked inside _run against the same S)
        assert torch.equal(torch.isnan(S1), torch.isnan(S2)) and np.array_equal(f1, f2)
        assert float((torch.nan_to_num(S1) - torch.nan_to_num(S2)).abs().max()) < 1e-9
        np.testing.assert_allclose(m1, m2, rtol=1e-12, atol=1e-12, equal_nan=True)
        if iz is None:
            assert 0.8 * n < status.sum() < n          # the odd columns fall back, the calibration is robust against them
        else:
            assert status.sum() == 0                  # ignore.zero = FALSE contradicts the calibration sample's rule
    gs.close()


def test_fused_entry_is_the_plain_route_when_it_does_not_apply(env):
    """few sets per column (the register-resident median kernel needs no help), few columns (nothing to calibrate on), the
    gather kernel, an unknown nnz: plaidhip_dev_spmm_csc_fused_f64 + ..._resume are exactly the plain pair"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g = 20000
    for m, n, kernel, fm in ((3000, 1100, "scatter", "on"), (24000, 300, "scatter", "on"), (24000, 1100, "gather", "on"),
                             (24000, 1100, "scatter", "auto"), (24000, 1100, "scatter", "off")):
        ctx.set_option("spmm_sparse_kernel", kernel)
        ctx.set_option("fused_medians", fm)
        Gp, Gi = sy.geneset_csc(g, m)
        gs = ctx.geneset(g, Gp, Gi)
        Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
        S1, S2, f1, f2, m1, m2, status, cal = _run(ctx, torch, dev, stream, gs, m, Xp, Xi, Xx, n)
        assert len(status) == 0
        if kernel == "gather":
            assert torch.equal(S1, S2)
        else:
            assert float((S1 - S2).abs().max()) < 1e-12
        assert np.array_equal(f1, f2) and np.allclose(m1, m2, rtol=0, atol=1e-12)
        gs.close()
    ctx.set_option("spmm_sparse_kernel", "scatter")
    ctx.set_option("fused_medians", "on")


def test_host_entries_take_the_fused_route_and_give_the_same_matrix(pinned_ctx):
    """plaid() and replaid.ssgsea() on a dgCMatrix through the host entry points (what R's .Call reaches): with the fused
    medians forced on and off the normalised score matrices are the same to the last bit (fixed-point sums, exact medians),
    and both meet the oracle"""
    import scipy.sparse as sp
    from oracle import plaid_oracle as po
    from plaid_amd import synth as sy
    g, m, n = 20000, 9000, 1100
    Gp, Gi = sy.geneset_csc(g, m)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    Xs = sp.csc_matrix((Xx, Xi, Xp), shape=(g, n))
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    outs = {}
    for mode in ("on", "off"):
        ctx = pinned_ctx(fused_medians=mode, spmm_sparse_kernel="scatter")
        outs[mode] = (ctx.plaid_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, "mean", True),
                      ctx.ssgsea_csc(Xs.indptr, Xs.indices, Xs.data, g, Gp, Gi, 0.25))
    assert np.array_equal(outs["on"][0], outs["off"][0]) and np.array_equal(outs["on"][1], outs["off"][1])
    np.testing.assert_allclose(outs["on"][0], po.plaid(Xs, rn, G, rn), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(outs["on"][1], po.replaid_ssgsea(Xs, rn, G, rn, alpha=0.25), rtol=1e-5, atol=1e-9)


def test_a_later_crossprod_into_the_same_matrix_invalidates_the_candidates(env):
    """fused crossprod of X1 into S, then a PLAIN crossprod of another X2 into the same S, then ..._resume(S): what the fused
    launch left behind describes a matrix that is gone -- the medians must be those of the S that is there"""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 20000, 9000, 1100
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    X1 = sy.sparse_columns(g, 0, n)
    X2 = sy.sparse_columns(g, 5000, 5000 + n)
    with torch.cuda.stream(stream):
        d1 = [torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (X1[0].astype(np.int32), X1[1].astype(np.int32), X1[2])]
        d2 = [torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (X2[0].astype(np.int32), X2[1].astype(np.int32), X2[2])]
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        fl = torch.zeros(4, dtype=torch.int32, device=dev)
        med_a = torch.zeros(n, dtype=torch.float64, device=dev)
        med_b = torch.zeros(n, dtype=torch.float64, device=dev)
        ctx.dev_spmm_csc_fused(gs, d1[0].data_ptr(), d1[1].data_ptr(), d1[2].data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0,
                               fl.data_ptr(), None, None, nnz=len(X1[2]))
        assert ctx.dev_fused_medians_info()[3]                       # a resume is pending
        fl.zero_()
        ctx.dev_spmm_csc(gs, d2[0].data_ptr(), d2[1].data_ptr(), d2[2].data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0,
                         fl.data_ptr(), None, nnz=len(X2[2]))
        assert not ctx.dev_fused_medians_info()[3]
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med_a.data_ptr(), fl.data_ptr())
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med_b.data_ptr(), fl.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(med_a, med_b)
    gs.close()


def test_resume_with_a_stale_token_never_uses_old_candidates(env):
    """the pending candidates of a fused crossprod used to be recognised by (S pointer, shape) alone -- an S that
    was re-written (or re-allocated at the same address) by somebody else would have got the OLD medians.  With the token:
    the matching token selects from the candidates; token 0 / a stale token / a discarded state run the standalone kernels
    on whatever S holds NOW."""
    torch, dev, stream, ctx = env
    from plaid_amd import synth as sy
    g, m, n = 20000, 9000, 1200
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    with torch.cuda.stream(stream):
        dp, di, dx = (torch.from_numpy(np.ascontiguousarray(a_)).to(dev) for a_ in (Xp.astype(np.int32), Xi.astype(np.int32), Xx))
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        fl = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        ref = torch.empty(n, dtype=torch.float64, device=dev)

        def fused():
            fl.zero_()
            return ctx.dev_spmm_csc_fused(gs, dp.data_ptr(), di.data_ptr(), dx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0,
                                          fl.data_ptr(), None, None, nnz=len(Xx))

        def status_sum():
            torch.cuda.synchronize()
            nf, p_status, _, _ = ctx.dev_fused_medians_info()
            st = np.zeros(max(nf, 1), dtype=np.int32)
            ctx.lib.plaidhip_memcpy_d2h(ctx.handle, st.ctypes.data_as(C.c_void_p), C.c_void_p(p_status), C.c_size_t(4 * nf))
            return int(st[:nf].sum())

        # 1. the matching token: candidates are used
        t1 = fused()
        assert t1 > 0 and ctx.dev_fused_medians_token() == t1
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), fl.data_ptr(), token=t1)
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, ref.data_ptr(), fl.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(med, ref) and ctx.dev_fused_medians_token() == 0 and status_sum() > 0.9 * n
        # 2. somebody else rewrites S at the same address; a stale token (the consumed one) and token 0 take the plain kernels
        t2 = fused()
        assert t2 > t1
        S.mul_(-3.0).add_(1.0)
        ctx.dev_col_medians(S.data_ptr(), m, m, n, False, ref.data_ptr(), fl.data_ptr())
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, False, med.data_ptr(), fl.data_ptr(), token=t1)
        torch.cuda.synchronize()
        assert torch.equal(med, ref), "a stale token must not select among the candidates of an older S"
        assert ctx.dev_fused_medians_token() == 0          # (and what was pending is dropped)
        # 3. token 0 == "nothing pending that I know of"
        fused()
        S.mul_(0.5)
        ctx.dev_col_medians(S.data_ptr(), m, m, n, False, ref.data_ptr(), fl.data_ptr())
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, False, med.data_ptr(), fl.data_ptr(), token=0)
        torch.cuda.synchronize()
        assert torch.equal(med, ref)
        # 4. explicit discard
        fused()
        ctx.dev_fused_medians_discard()
        assert ctx.dev_fused_medians_token() == 0
    gs.close()


def test_phase_engine_fuses_only_for_a_normalising_caller(env):
    """sharded.HipPhaseEngine.spmm_csc(normalize=False) takes the plain crossprod and leaves nothing pending; with
    normalize=True, medians() resumes only for the tensor that launch returned"""
    torch, dev, stream, ctx = env
    from plaid_amd import sharded, synth as sy
    g, m, n = 20000, 9000, 1100
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    Xp, Xi, Xx = sy.sparse_columns(g, 0, n)
    with torch.cuda.stream(stream):
        import scipy.sparse as sps
        X = sharded.CscShard.from_scipy(sps.csc_matrix((Xx, Xi, Xp), shape=(g, n)), 0, n, dev)
        eng = sharded.HipPhaseEngine(ctx, gs, dev)
        fl = eng.new_flags()
        S0 = eng.spmm_csc(X, flags=fl, normalize=False)
        assert ctx.dev_fused_medians_token() == 0
        fl = eng.new_flags()
        S1 = eng.spmm_csc(X, flags=fl, normalize=True)
        assert ctx.dev_fused_medians_token() > 0 and torch.equal(S0, S1)
        # medians of ANOTHER tensor first: plain kernels, the pending state is dropped, not misapplied
        other = (S1 * 2.0 - 1.0).contiguous()
        fo = eng.new_flags()
        ctx.dev_minflags(other.data_ptr(), other.numel(), fo.data_ptr())
        med_o, _ = eng.medians(other, fo)
        ref = torch.empty(n, dtype=torch.float64, device=dev)
        ctx.dev_col_medians(other.data_ptr(), m, m, n, None, ref.data_ptr(), fo.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(med_o, ref) and ctx.dev_fused_medians_token() == 0
        med1, _ = eng.medians(S1, fl)
        ctx.dev_col_medians(S1.data_ptr(), m, m, n, None, ref.data_ptr(), fl.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(med1, ref)
    gs.close()
