#!/usr/bin/python3
"""sharded.gather_scores(to="host") with several ranks on ONE GPU (gloo for the handshakes; the data path is device ->
pinned slab -> this rank's /dev/shm file, no collective): rate and content.   On the GPU box:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/gather_host_check.py 60000"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from plaid_amd import sharded
    rows_per_rank = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    m = 50001                                           # (an odd row length: block boundaries fall inside pages)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    n_total = rows_per_rank * world - 3                 # (uneven: the last rank is short)
    lo, hi = sharded.shard_bounds(n_total, world, rank)
    S = (torch.arange(lo, hi, device=dev, dtype=torch.float64)[:, None] * 1e-3 +
         torch.arange(m, device=dev, dtype=torch.float64)[None, :] * 1e-9)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    full = sharded.gather_scores(S, n_total, dst=0, to="host")
    dt = time.perf_counter() - t0
    if rank == 0:
        gb = n_total * m * 8 / 1e9
        print(f"{world} ranks: {gb:.1f} GB in {dt:.2f} s = {gb / dt:.1f} GB/s", flush=True)
        for r in (0, 1, hi - 1, hi, n_total // 2, n_total - 1):
            if r < n_total:
                exp = r * 1e-3 + np.arange(m) * 1e-9
                assert np.array_equal(full[r], exp), r
        print("content ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
