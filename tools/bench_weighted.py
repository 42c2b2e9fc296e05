"""Times the general weighted crossprod (kernels_wspmm.hip) on the C2 shape beside the scheduled membership kernel.
    python3 tools/bench_weighted.py [--genes 20000 --samples 10000 --sets 5000 --iters 5]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--sets", type=int, default=5000)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import torch
    import plaid_amd
    from plaid_amd import synth as sy
    g, n, m = a.genes, a.samples, a.sets
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    Gp, Gi = sy.geneset_csc(g, m)
    gs = ctx.geneset(g, Gp, Gi)
    rng = np.random.default_rng(0)
    Wx = rng.normal(size=len(Gi))
    ldx = g + (g & 1)
    with torch.cuda.stream(stream):
        X = torch.rand((n, ldx), dtype=torch.float64, device=dev)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        S2 = torch.empty((n, m), dtype=torch.float64, device=dev)
        dWp = torch.from_numpy(Gp.astype(np.int32)).to(dev)
        dWi = torch.from_numpy(Gi.astype(np.int32)).to(dev)
        dWx = torch.from_numpy(Wx).to(dev)
        dW1 = torch.ones(len(Gi), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(a.iters):
                fn()
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    t_w = timed(lambda: ctx.dev_crossprod_weighted(dWp.data_ptr(), dWi.data_ptr(), dWx.data_ptr(), g, m, X.data_ptr(), ldx, n,
                                                   S.data_ptr(), m))
    t_m = timed(lambda: ctx.dev_spmm_dense(gs, X.data_ptr(), ldx, n, S2.data_ptr(), m, stat="sum"))
    # same numbers with unit weights
    ctx.dev_crossprod_weighted(dWp.data_ptr(), dWi.data_ptr(), dW1.data_ptr(), g, m, X.data_ptr(), ldx, n, S.data_ptr(), m)
    torch.cuda.synchronize()
    err = float(((S - S2).abs() / S2.abs().clamp_min(1e-300)).max())
    z = len(Gi)
    print(f"genes {g} samples {n} sets {m} memberships {z}")
    print(f"weighted kernel   {t_w:8.3f} ms   {m * n / t_w / 1e6:8.2f} Gscores/s   x stream {z * 12 * n / t_w / 1e6:8.1f} GB/s from L2")
    print(f"membership kernel {t_m:8.3f} ms   {m * n / t_m / 1e6:8.2f} Gscores/s")
    print(f"unit weights: max rel diff between the two {err:.2e}")


if __name__ == "__main__":
    main()
