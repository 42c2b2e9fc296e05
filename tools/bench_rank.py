#!/usr/bin/python3
"""Column-rank kernels alone (GPU): bucket ranker vs sorting network at config-4 shape, tied and tie-free data, with
and without the fused power; with the tools/ build (PLAIDHIP_LIB=plaid_amd/csrc/libplaidhip_diag.so) also the share of
every phase of the bucket kernel (in-kernel stamps of wave 0).   python tools/bench_rank.py [--genes 20000] [--cols 4096]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    g, n = a.genes, a.cols
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    X = torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen) * 2 + 8
    data = {"tie-free N(8,2)": X, "rounded to 0.1": torch.round(X * 10) / 10,
            "95% zeros": torch.where(torch.rand((n, g), device=dev, generator=gen) < 0.95, torch.zeros_like(X), torch.round(X * 10) / 10)}
    R = torch.empty_like(X)
    colmax = torch.empty(n, dtype=torch.float64, device=dev)
    diag = hasattr(ctx.lib, "plaidhip_debug_set_rank_stamps")
    dbg = torch.zeros((n, 8), dtype=torch.int64, device=dev)
    if diag:
        ctx.lib.plaidhip_debug_set_rank_stamps.argtypes = [ctypes.c_void_p]
        ctx.lib.plaidhip_debug_set_rank_stamps(dbg.data_ptr())
    for name, Xd in data.items():
        for kern in ("bucket", "bucket512", "network"):
            for power in (1.0, 1.25):
                ctx.set_option("rank_kernel", kern)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
                with torch.cuda.stream(stream):
                    ctx.dev_colranks_dense(Xd.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, power, colmax.data_ptr())
                    for e0, e1 in ev:
                        e0.record(stream)
                        ctx.dev_colranks_dense(Xd.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, power, colmax.data_ptr())
                        e1.record(stream)
                torch.cuda.synchronize()
                ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
                line = f"{name:18s} {kern:8s} power {power:4.2f}: {ms:8.3f} ms  {g * n / ms / 1e6:8.2f} Gkeys/s  {16.0 * g * n / ms / 1e6:7.1f} GB/s algorithmic"
                if diag and kern.startswith("bucket"):
                    t = dbg.cpu().numpy().astype(np.float64).sum(axis=0)
                    line += "  phases% " + " ".join(f"{100 * v / t.sum():.0f}" for v in t[:6])
                print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
