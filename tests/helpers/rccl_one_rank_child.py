"""Child process of tests/test_gpu_rccl.py: the sample-sharded host path (plaid_amd/sharded.py) over RCCL with ONE rank.

A 1-GPU box cannot run RCCL with two ranks, but it can run every RCCL call the 8-GPU job makes: `init_process_group("nccl",
device_id=...)`, the int32 MAX and fp64 SUM all-reduces on device tensors (sharded._collective runs them for a group of one
rank too), the broadcast of the gather verdict, and `gather_scores` to the device and to the host.  The results must be
bit-identical to the same calls made before the process group existed (no collective at all) -- R/plaid.R:107, :634-642 shard
by sample column; the three scalars of :251, :557, :572 are what the all-reduces carry.  Prints one JSON line; rc 0 = equal."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    import plaid_amd
    from plaid_amd import sharded, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    g, n, m = 2000, 768, 7000                       # more than 6,144 sets: the two-chunk median kernels
    Gp, Gi = synth.geneset_csc(g, m, kmin=5, kmax=120)
    gs = ctx.geneset(g, Gp, Gi)
    rng = np.random.default_rng(11)
    Xd = np.round(rng.normal(8.0, 2.0, size=(n, g)), 1)
    dens = sp.random(g, n, density=0.06, format="csc", random_state=5, data_rvs=lambda k: np.round(rng.gamma(2.0, 1.0, k), 1) + 0.1)
    out = {}
    with torch.cuda.stream(stream):
        eng = sharded.HipPhaseEngine(ctx, gs, dev)
        X = torch.from_numpy(Xd).to(dev)
        Xs = sharded.CscShard.from_scipy(dens, 0, n, dev)

        def run_all():
            a = sharded.sharded_plaid(eng, X)
            b = sharded.sharded_ssgsea(eng, X, alpha=0.25)
            c = sharded.sharded_ssgsea_csc(eng, Xs, alpha=0.25)
            d = sharded.sharded_plaid_csc(eng, Xs)
            torch.cuda.synchronize()
            return [t.clone() for t in (a, b, c, d)]

        assert not dist.is_initialized()
        ref = run_all()                               # no process group: no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        out["backend"] = dist.get_backend()
        out["world"] = dist.get_world_size()
        got = run_all()                               # the same calls, every scalar through an RCCL all-reduce
        names = ["plaid_dense", "ssgsea_dense", "ssgsea_csc", "plaid_csc"]
        out["equal"] = {nm: bool(torch.equal(r, t)) for nm, r, t in zip(names, ref, got)}
        # explicit collectives on device tensors, as the N-GPU job issues them
        f = torch.tensor([1, 0, 3, 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        r2 = torch.tensor([2.5, 7.0], dtype=torch.float64, device=dev)
        dist.all_reduce(r2, op=dist.ReduceOp.SUM)
        out["allreduce_ok"] = f.tolist() == [1, 0, 3, 0] and r2.tolist() == [2.5, 7.0]
        full_d = sharded.gather_scores(got[2], n, to="device")
        full_f = sharded.gather_scores(got[2], n, to="device", dtype=torch.float32)
        full_h = sharded.gather_scores(got[2], n, to="host")
        torch.cuda.synchronize()
        out["gather_device_equal"] = bool(torch.equal(full_d, ref[2]))
        out["gather_device_fp32_equal"] = bool(torch.equal(full_f, ref[2].to(torch.float32)))
        out["gather_host_equal"] = bool(np.array_equal(np.asarray(full_h), ref[2].cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    ok = (out["backend"] == "nccl" and out["world"] == 1 and all(out["equal"].values()) and out["allreduce_ok"]
          and out["gather_device_equal"] and out["gather_device_fp32_equal"] and out["gather_host_equal"])
    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
