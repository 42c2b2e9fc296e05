#!/usr/bin/env python
"""Headline benchmark: sample x geneset scores/sec of plaid() at 20k genes on MI355X.

A "step" is one pass of the hot path over one resident batch: S = G^T X (sparse 0/1
membership x dense expression, 1/|set| scaling) followed by normalize_medians()
(R/plaid.R:60-87).  Workload at N=1 = BASELINE.json configs[1] (C2): synthetic dense
20,000 genes x 10,000 samples x 5,000 gene sets, inputs resident in HBM.  With --gpus N
(launched by torch.distributed.run, one rank per GPU) every rank holds its own 10,000-sample
shard (weak scaling); the only data-path collectives are the two scalar all-reduces
normalize_medians needs across shards (min(x)==0 flags, mean of medians).  The optional
gather of the score shards to rank 0 is timed separately and never part of `value`.

Prints ONE JSON line (rank 0).  `roofline` is the SpMM kernel's algorithmic HBM bytes /
its HIP-event time; `cpu_baseline` times the plain-C oracle (the reference is R and cannot
run here) on a bounded column sample of the same workload, one core.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=10000, help="samples per GPU")
    ap.add_argument("--sets", type=int, default=5000)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="columns timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-mixed", action="store_true", help="skip the secondary mixed-precision (fp32-staged) measurement")
    return ap.parse_args()


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import plaid_amd
    from plaid_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on 1 GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    g, n, m = a.genes, a.samples, a.sets
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(local_rank, stream.cuda_stream)

    # ---- synthetic inputs, resident in HBM before the timed region -----------------------
    Gp, Gi = synth.geneset_csc(g, m)
    z = int(Gp[-1])
    gs = ctx.geneset(g, Gp, Gi)
    X = torch.empty((n, g), dtype=torch.float64, device=dev)      # row-major (n, g) == R's g x n
    col0 = rank * n                                               # this rank's sample shard
    for j0 in range(0, n, 1024):
        j1 = min(n, j0 + 1024)
        blk = synth.dense_columns(g, col0 + j0, col0 + j1)        # (g, b) Fortran
        X[j0:j1].copy_(torch.from_numpy(np.ascontiguousarray(blk.T)))
    S = torch.empty((n, m), dtype=torch.float64, device=dev)      # == m x n column-major
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    med = torch.empty(n, dtype=torch.float64, device=dev)
    red = torch.zeros(4, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(k=None):
        with torch.cuda.stream(stream):
            flags.zero_()
            if k is not None:
                ev[k][0].record(stream)
            ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
            if k is not None:
                ev[k][1].record(stream)
            if use_dist:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)                  # min(x)==0 over all shards
            ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
            ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
            if use_dist:
                dist.all_reduce(red, op=dist.ReduceOp.SUM)                    # mean(medx) over all shards
            ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
            if k is not None:
                ev[k][2].record(stream)

    for _ in range(a.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_step = 1e3 * elapsed / a.steps
    spmm_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    norm_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    scores = float(world) * n * m
    value = scores / (elapsed / a.steps)

    # ---- secondary, reported beside the headline, never in `value`: the opt-in mixed-precision crossprod
    #      (sample columns staged as fp32 in LDS, sums fp64), same workload, same step ---------------------
    mixed = None
    if not a.no_mixed and world == 1:   # N = 1 only, like cpu_baseline: no collective may depend on an optional block
        try:
            ctx.set_precision("mixed")
            msteps = max(3, min(a.steps, 10))
            mev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(msteps)]
            step()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            for k in range(msteps):
                with torch.cuda.stream(stream):
                    mev[k][0].record(stream)
                step()
                with torch.cuda.stream(stream):
                    mev[k][1].record(stream)
            torch.cuda.synchronize()
            tm = (time.perf_counter() - tm) / msteps
            # the SpMM is the first kernel of a step: time it alone once more with events around it
            sp_ev = []
            for _ in range(msteps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, None)
                    e1.record(stream)
                sp_ev.append((e0, e1))
            torch.cuda.synchronize()
            m_spmm = float(np.mean([e0.elapsed_time(e1) for e0, e1 in sp_ev]))
            mixed = {"ms_per_step": round(1e3 * tm, 4), "scores_per_s": round(n * m / tm, 1), "spmm_ms": round(m_spmm, 4),
                     "spmm_algorithmic_GBps": round((g * n * 8 + 4 * z + 4 * (m + 1) + m * n * 8) / (m_spmm * 1e-3) / 1e9, 1),
                     "note": "opt-in (plaidhip_set_precision MIXED): inputs rounded to fp32 in LDS, fp64 sums; rank 0 only, "
                             "not part of value"}
        except Exception as exc:  # pragma: no cover
            mixed = {"error": str(exc)[:200]}
        finally:
            ctx.set_precision("f64")

    # ---- roofline of the dominant kernel (SpMM): algorithmic bytes per launch --------------
    # g*n*b_X + (4 z + 4 (m+1)) + m*n*b_S with b = 8 (fp64 in, fp64 out)
    alg_bytes = g * n * 8 + 4 * z + 4 * (m + 1) + m * n * 8
    # dense X with 16-byte aligned columns takes the two-columns-per-pass kernel (kernels_spmm.hip)
    spmm_kernel = "spmm_colpair_f64" if (g % 2 == 0 and os.environ.get("PLAIDHIP_SPMM_KERNEL") != "single") \
        else "spmm_colgather_f64"
    achieved = alg_bytes / (spmm_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{spmm_kernel}/{g}x{n}x{m}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": spmm_kernel, "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "algorithmic_bytes": alg_bytes, "kernel_ms": round(spmm_ms, 4)}

    # ---- optional gather of the score shards to rank 0 (reported, never in `value`) --------
    gather = None
    if use_dist and not a.no_gather:
        try:
            from plaid_amd import sharded
            dist.barrier()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            full = sharded.gather_scores(S, world * n, dst=0)   # grouped peer->root irecv/isend
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter() - tg
            del full
            nbytes = (world - 1) * S.numel() * 8
            gather = {"ms": round(1e3 * tg, 3), "GB/s_into_root": round(nbytes / tg / 1e9, 1)}
        except Exception as exc:  # pragma: no cover
            gather = {"error": str(exc)[:200]}

    # ---- CPU baseline + parity spot-check on a bounded sample (rank 0, N = 1 only) ---------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        from oracle import c_oracle
        nc = min(a.cpu_sample, n)
        Xh = synth.dense_columns(g, 0, nc)
        t1 = time.perf_counter()
        Sraw = c_oracle.plaid_dense(Xh, Gp, Gi, "mean", False)
        t2 = time.perf_counter()
        c_oracle.normalize_medians(Sraw)
        t3 = time.perf_counter()
        cpu = {"value": round(m * nc / (t3 - t1), 1), "unit": "scores/s", "cores": 1, "kind": "port",
               "sample": f"first {nc} of {n} sample columns x {m} sets, plain-C oracle (oracle/plaid_oracle.c): "
                         f"crossprod {t2 - t1:.2f} s + normalize_medians {t3 - t2:.2f} s",
               "cpu_count": os.cpu_count()}
        # checker: the GPU's un-normalised scores for the same columns
        with torch.cuda.stream(stream):
            ctx.dev_spmm_dense(gs, X.data_ptr(), g, nc, S.data_ptr(), m, "mean", 1.0, 0.0, None)
        torch.cuda.synchronize()
        Sg = S[:nc].cpu().numpy().T
        parity = {"max_rel_err_vs_oracle": float(np.max(np.abs(Sg - Sraw) / np.maximum(np.abs(Sraw), 1e-300))),
                  "columns": nc}
        if mixed is not None and "error" not in mixed:
            ctx.set_precision("mixed")
            with torch.cuda.stream(stream):
                ctx.dev_spmm_dense(gs, X.data_ptr(), g, nc, S.data_ptr(), m, "mean", 1.0, 0.0, None)
            torch.cuda.synchronize()
            ctx.set_precision("f64")
            Sm = S[:nc].cpu().numpy().T
            mixed["max_rel_err_vs_oracle"] = float(np.max(np.abs(Sm - Sraw) / np.maximum(np.abs(Sraw), 1e-300)))

    if rank == 0:
        out = {
            "metric": "sample x geneset scores/sec at 20k genes (plaid(): crossprod + median normalisation)",
            "value": round(value, 1), "unit": "scores/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2 dense plaid(): {g} genes x {n} samples/GPU x {m} gene sets "
                                   f"(z={z} memberships), inputs resident in HBM",
                       "genes": g, "samples_per_gpu": n, "sets": m, "memberships": z,
                       "parallelism": f"sample-shard x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
            "phases_ms": {"spmm": round(spmm_ms, 4), "normalize_medians": round(norm_ms, 4)},
            "parity": parity, "gather": gather, "mixed_precision": mixed,
        }
        print(json.dumps(out))
    gs.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
