"""Full-size launches against the oracle (`pytest -m gpu`): results whose element offsets cross 2^31 -- the limit
`chunked_crossprod` chunks for (R/plaid.R:103-104) -- and the per-GPU shard of BASELINE config 5.

A result of 2e9 ... 6e9 scores cannot be recomputed on the CPU; single columns can (oracle/fullsize.py): the probe
columns are the first ones, the ones either side of element offset 2^31, and the LAST ones of the very launch under
test, so every 64-bit offset, the persistent grid's late columns and the reused per-wavefront state are covered."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_ctx():
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    yield torch, dev, stream, ctx
    ctx.close()


def _np_f(t):
    return np.asfortranarray(t.cpu().numpy().T)


def test_plaid_dense_result_offsets_cross_2_31(torch_ctx):
    """plaid() on dense 20,000 genes x 43,100 samples x 50,000 sets: S holds 2.155e9 doubles (17.2 GB), so the last
    151 columns start past element 2^31.  Probe columns: first 8, 8 around element 2^31, last 8."""
    from oracle import c_oracle, fullsize
    from plaid_amd import synth as sy
    torch, dev, stream, ctx = torch_ctx
    g, n, m = 20000, 43100, 50000
    assert m * n > 2**31
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    with torch.cuda.stream(stream):
        X = torch.empty((n, g), dtype=torch.float64, device=dev)
        for j0 in range(0, n, 8192):
            j1 = min(n, j0 + 8192)
            X[j0:j1] = torch.randn((j1 - j0, g), dtype=torch.float64, device=dev, generator=gen) * 2.0 + 8.0
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        red = torch.zeros(2, dtype=torch.float64, device=dev)
        ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
    torch.cuda.synchronize()
    cols, crosses = fullsize.probe_columns(n, m, 8)
    assert crosses and cols[-1] == n - 1 and int(cols[-8]) * m > 2**31
    idx = torch.as_tensor(cols, device=dev)
    raw_g = _np_f(S.index_select(0, idx))
    minmax = (float(S.min().item()), bool((S == 0).any().item()))
    with torch.cuda.stream(stream):
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
        ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
        ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
    torch.cuda.synchronize()
    Xc = _np_f(X.index_select(0, idx))
    raw_o = c_oracle.crossprod_dense(Xc, Gp, Gi, "mean", fullsize._threads())
    np.testing.assert_allclose(raw_g, raw_o, rtol=1e-12, atol=0)          # summation order only
    res = fullsize.check_normalised(raw_o, _np_f(S.index_select(0, idx)), med.cpu().numpy(), cols,
                                    red.cpu().numpy(), flags.cpu().numpy(), minmax)
    assert res["max_rel_err_vs_oracle"] < 1e-10
    gs.close()
    del X, S
    torch.cuda.empty_cache()


def test_config5_shard_ssgsea_csc_125k_cells(torch_ctx):
    """BASELINE config 5 per GPU: replaid.ssgsea(alpha = 0.25) on a sparse 20,000 x 125,000-cell shard x 50,000 sets
    (6.25e9 scores, 50 GB).  First 8 cells, 8 around element 2^31, last 8 against the oracle -- ranks bit-exact, max(rX)
    and mean(medx) verified from the device's per-cell vectors -- plus idempotence of the median normalisation over
    the whole shard."""
    from oracle import fullsize
    from plaid_amd import synth as sy
    torch, dev, stream, ctx = torch_ctx
    g, n, m, alpha = 20000, 125000, 50000, 0.25
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    with torch.cuda.stream(stream):
        Xp, Xi, Xx, nnz, max_nnz = sy.device_sparse_cells(torch, dev, g, n, 99)
        Rx = torch.empty_like(Xx)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        colmax = torch.zeros(n, dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        med2 = torch.empty(n, dtype=torch.float64, device=dev)
        small = torch.zeros(8, dtype=torch.float64, device=dev)
        red, gmax = small[0:2], small[2:3]
        ctx.dev_colranks_csc(Xp.data_ptr(), Xx.data_ptr(), n, max_nnz, Rx.data_ptr(), "average", False, 1.0 + alpha,
                             colmax.data_ptr())
        ctx.dev_max(colmax.data_ptr(), n, gmax.data_ptr())
        ctx.dev_spmm_csc(gs, Xp.data_ptr(), Xi.data_ptr(), Rx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, -0.5,
                         flags.data_ptr(), gmax.data_ptr(), nnz=nnz)
        ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
        ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
        ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
        # second normalisation pass over the whole shard: every median is now the common value
        ctx.dev_col_medians(S.data_ptr(), m, m, n, False, med2.data_ptr(), None)
    torch.cuda.synchronize()
    cols, crosses = fullsize.probe_columns(n, m, 8)
    assert crosses
    ph = Xp.cpu().numpy()
    parts, rparts = [], []
    for lo, hi in fullsize.contiguous_runs(cols):
        q0, q1 = int(ph[lo]), int(ph[hi])
        parts.append((ph[lo:hi + 1], Xi[q0:q1].cpu().numpy(), Xx[q0:q1].cpu().numpy()))
        rparts.append(Rx[q0:q1].cpu().numpy())
    sp_, si_, sx_ = fullsize.sub_csc(parts)
    rx_g = np.concatenate(rparts)
    colmax_h, gmax_h = colmax.cpu().numpy(), float(gmax.cpu().numpy()[0])
    assert gmax_h == colmax_h.max()
    r_o, raw_o = fullsize.ssgsea_csc_raw(sp_, si_, sx_, g, Gp, Gi, alpha, gmax_h)
    assert np.array_equal(fullsize.ranks_from_powered(rx_g, 1.0 + alpha), r_o)
    idx = torch.as_tensor(cols, device=dev)
    res = fullsize.check_normalised(raw_o, _np_f(S.index_select(0, idx)), med.cpu().numpy(), cols, red.cpu().numpy(),
                                    flags.cpu().numpy(), None)
    assert res["max_abs_err_vs_oracle"] < 1e-12
    m2 = med2.cpu().numpy()
    mean1 = float(np.mean(med.cpu().numpy()))
    np.testing.assert_allclose(m2, np.full(n, mean1), rtol=0, atol=1e-12)
    gs.close()
    del S, Xx, Xi, Rx
    torch.cuda.empty_cache()
