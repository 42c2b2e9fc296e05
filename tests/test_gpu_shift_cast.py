"""plaidhip_dev_shift_columns_cast_f32 (`pytest -m gpu`): the last step of normalize_medians (R/plaid.R:572) fused with the
fp64 -> fp32 cast of a sample-sharded gather (config 5's result must be fp32 to fit the root) -- bit-identical to the sweep
followed by a conversion, for aligned and misaligned columns, odd lengths and empty shapes; and through
sharded.gather_scores(dtype=float32, shift=...) on one rank."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(37, 5000), (16, 5001), (3, 1), (5, 7), (0, 64), (9, 50001)])
def test_shift_cast_equals_shift_then_cast(n, m):
    import torch
    import plaid_amd
    from plaid_amd import sharded
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    rng = np.random.default_rng(5)
    S_h = rng.normal(0.0, 1.0, size=(n, m)) * 10.0 ** rng.integers(-6, 4, size=(n, 1))
    med_h = np.median(S_h, axis=1) if m else np.zeros(n)
    red_h = np.array([med_h.sum(), float(n)]) if n else np.array([0.0, 1.0])
    exp = ((S_h - med_h[:, None]) + (red_h[0] / red_h[1])).astype(np.float32)
    with torch.cuda.stream(stream):
        S = torch.from_numpy(S_h).to(dev)
        med = torch.from_numpy(med_h).to(dev)
        red = torch.from_numpy(red_h).to(dev)
        out = torch.full((n, m), float("nan"), dtype=torch.float32, device=dev)
        if n and m:
            ctx.dev_shift_columns_cast_f32(S.data_ptr(), m, m, n, med.data_ptr(), out.data_ptr(), m, 0.0, red.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), exp, equal_nan=False) or n * m == 0
        assert np.array_equal(S.cpu().numpy(), S_h)                       # the source is left alone
        if n and m:
            # the same through the gather on one rank (no process group): slabs of 4 rows
            Gp = np.arange(0, 4, dtype=np.int32)                          # any gene-set handle: the engine only needs a context here
            gs = ctx.geneset(8, Gp, np.arange(3, dtype=np.int32))
            eng = sharded.HipPhaseEngine(ctx, gs, dev)
            full = sharded.gather_scores(S, n, to="device", dtype=torch.float32, chunk_rows=4, shift=(eng, med, red))
            torch.cuda.synchronize()
            assert np.array_equal(full.cpu().numpy(), exp)
            # and against the library's own sweep followed by torch's cast
            S2 = S.clone()
            ctx.dev_shift_columns(S2.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(S2.to(torch.float32), full)
            gs.close()
    ctx.close()
