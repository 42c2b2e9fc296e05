"""A/B on one GPU: what a shard does to its score block before the gather to an fp32 root matrix (config 5).
  a) shift_columns in place (read + write fp64) then a cast of the block (read fp64, write fp32)        -- rounds 2-5
  b) shift_columns_cast_f32: one read of the fp64 block, one fp32 write                                   -- round 6
usage: python3 tools/bench_shift_cast.py [cells] [sets]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plaid_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ctx = plaid_amd.Context(0, stream.cuda_stream)
with torch.cuda.stream(stream):
    S = torch.randn((n, m), dtype=torch.float64, device=dev)
    med = torch.randn(n, dtype=torch.float64, device=dev) * 1e-3
    red = torch.tensor([0.25, float(n)], dtype=torch.float64, device=dev)
    out = torch.empty((n, m), dtype=torch.float32, device=dev)

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e
    for rep in range(4):
        e0 = ev()
        ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
        e1 = ev()
        out.copy_(S)
        e2 = ev()
        ctx.dev_shift_columns_cast_f32(S.data_ptr(), m, m, n, med.data_ptr(), out.data_ptr(), m, 0.0, red.data_ptr())
        e3 = ev()
        torch.cuda.synchronize()
        if rep:
            a, b, c = e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)
            gb = n * m / 1e9
            print(f"{n} x {m}: shift in place {a:.3f} ms + cast {b:.3f} ms = {a + b:.3f} ms ({28e3 * gb / (a + b):.0f} GB/s over 28 B per score)   "
                  f"shift_columns_cast_f32 {c:.3f} ms ({12e3 * gb / c:.0f} GB/s over 12 B per score)   ratio {(a + b) / c:.2f}x")
