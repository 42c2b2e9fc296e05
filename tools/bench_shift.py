#!/usr/bin/python3
"""shift_columns alone (GPU): (x - med[col]) + mean at the C2 and C3 column lengths.   python tools/bench_shift.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    shapes = ((10000, 5000), (8192, 50000), (8192, 49999))
    if len(sys.argv) > 1 and sys.argv[1] == "big":   # + the reference's published shape and config 3's full size (40 GB)
        shapes += ((10000, 61459), (100000, 50000))
    for n, m in shapes:
        with torch.cuda.stream(stream):
            S = torch.empty((n, m), dtype=torch.float64, device=dev)
            S.zero_()
            med = torch.randn(n, dtype=torch.float64, device=dev)
            red = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for a, b in ev:
                a.record(stream)
                ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
                b.record(stream)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
        print(f"shift {n} x {m}: median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f} ms -> {16.0 * n * m / ms[len(ms) // 2] / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
