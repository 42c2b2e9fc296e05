"""Column ranks of dense columns longer than the LDS-resident rank kernel takes (20,352 rows): the value-partitioned route
against the sorting network on a global scratch (plaidhip_set_option rank_kernel = network forces the old route).
    python3 tools/bench_rank_long.py [--cols 2048]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cols", type=int, default=2048)
    a = ap.parse_args()
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    n = a.cols
    for g in (20352, 25000, 33538, 60000, 120000):
        gen = torch.Generator(device=dev)
        gen.manual_seed(g)
        X = torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen) * 2 + 8
        data = {"tie-free": X, "94% zeros": torch.where(torch.rand((n, g), device=dev, generator=gen) < 0.94, torch.zeros_like(X),
                                                        torch.round(X * 10) / 10)}
        R = torch.empty_like(X)
        for name, Xd in data.items():
            line = f"{g:7d} rows x {n} columns, {name:9s}:"
            for kern in ("auto", "network"):
                ctx.set_option("rank_kernel", kern)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
                with torch.cuda.stream(stream):
                    ctx.dev_colranks_dense(Xd.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, 1.0, None)
                    for e0, e1 in ev:
                        e0.record(stream)
                        ctx.dev_colranks_dense(Xd.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, 1.0, None)
                        e1.record(stream)
                torch.cuda.synchronize()
                ms = min(e0.elapsed_time(e1) for e0, e1 in ev)
                line += f"  {kern} {ms:8.3f} ms ({g * n / ms / 1e6:6.2f} Gkeys/s)"
            print(line, flush=True)
        del X, data, R
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
