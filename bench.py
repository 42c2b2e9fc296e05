#!/usr/bin/python3
"""Headline benchmark: sample x geneset scores/sec of the plaid hot path at 20k genes on MI355X.

A "step" is one pass of the hot path over one resident batch of synthetic input.

  headline (`value`, every N)   BASELINE.json configs[1] = C2: plaid() on dense 20,000 genes x 10,000 samples per
                                GPU x 5,000 gene sets: S = G^T X (1/|set| scaling) + normalize_medians()
                                (R/plaid.R:60-87).  With --gpus N every rank holds its own 10,000-sample shard
                                (weak scaling); the data-path collectives are the two scalar all-reduces
                                normalize_medians needs across shards.
  "c3" block (N = 1)            configs[2]: replaid.ssgsea(alpha = 0.25) on SPARSE 20,000 x 100,000 cells (5 % stored)
                                x 50,000 sets, full size: sparse_colranks -> max(rX) -> scatter crossprod -> medians.
  "c4" block (N = 1)            configs[3]: replaid.ssgsea(alpha = 0.25) on DENSE 20,000 x 50,000 x 50,000, full size:
                                colranks (bucket ranker, fused power) -> crossprod -> medians.  (The reference multiplies
                                the sparse G here too, R/plaid.R:253 -> :80; the dense MFMA contraction is priced in
                                DESIGN.md and tools/.)
  "c5_shard" block (N > 1)      configs[4] per GPU: the c3 pipeline on a 125,000-cell CSC shard per rank with the
                                three scalar all-reduces of plaid_amd/sharded.py (sharded_ssgsea_csc).

Prints ONE COMPACT JSON line (rank 0, < 4 KB: `compact_line`), the last thing on stdout; the full record (every block's
roofline / cpu_baseline / parity entries) goes to bench_detail.json beside this file (BENCH_DETAIL=path moves it).  `roofline` = the
dominant kernel's algorithmic HBM bytes / its HIP-event time (`frac`), with the LDS-return roof the fp64 gather kernels
actually sit on beside it (`lds_frac`); `cpu_baseline` times the plain-C oracle (the reference is R and cannot run here)
on a bounded column sample of the same workload: one core (reference-faithful) and all cores (OpenMP over columns).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (torch.distributed.run as a CHILD
process, before anything here touches the GPU) and forwards the child's line and exit status.
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

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LDS_PEAK_GBPS = 157286.4    # 256 CUs x 256 B/clk x 2.4 GHz (MI355X_MICROARCH.md, LDS: ds_read_b64/b128)
FP64_PEAK_TFLOPS = 78.6     # vector FP64, FMA counted as 2 flop
METRIC = "sample x geneset scores/sec at 20k genes (plaid(): crossprod + median normalisation)"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=10000, help="C2 samples per GPU")
    ap.add_argument("--sets", type=int, default=5000, help="C2 gene sets")
    ap.add_argument("--config", default="all", choices=["all", "c2", "c3", "c4", "ref", "c3real"],
                    help="N = 1: which blocks to run next to the C2 headline (default: all)")
    ap.add_argument("--profile", action="store_true",
                    help="profiling run (tools/profile_round.sh): ONLY the block --config names, no C2 headline, no pre-heat, "
                         "so that a rocprofv3 kernel-stats file holds one launch shape per kernel row")
    ap.add_argument("--ref-samples", type=int, default=10000, help="samples of the ref_shape block (the reference's published runs: 10,000)")
    ap.add_argument("--c3-cells", type=int, default=100000)
    ap.add_argument("--c4-samples", type=int, default=50000)
    ap.add_argument("--c5-cells-per-gpu", type=int, default=125000)
    ap.add_argument("--big-sets", type=int, default=50000, help="gene sets of the c3 / c4 / c5 blocks")
    ap.add_argument("--block-steps", type=int, default=5, help="timed steps of the c3 / c4 / c5 blocks")
    ap.add_argument("--preheat-steps", type=int, default=60,
                    help="untimed C2 steps before the --warmup steps, so that the timed steps run at the sustained clocks (0: none)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="C2 columns timed on the CPU oracle (0 = skip all CPU legs)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--host-gather", action="store_true", help="N > 1: also time the host (/dev/shm) gather of the C5 shard's score matrix")
    ap.add_argument("--no-mixed", action="store_true", help="skip the secondary mixed-precision (fp32-staged) measurement")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: rendezvous (gloo), the max-over-ranks reduction and the output line only (CPU test of the launcher)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------- the line the driver parses
LINE_LIMIT = 4096   # bytes: the driver stopped parsing the line when it grew to ~23 KB (BENCH_r04.json: parsed = null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _parity_ok(p):
    """one boolean out of a parity report: a report without an explicit verdict exists only because every assertion of
    the checker held (oracle/fullsize.py raises otherwise; the except arms store ok = False)"""
    if not isinstance(p, dict):
        return None
    if "ok" in p:
        return bool(p["ok"])
    return "error" not in p


def _block_summary(b):
    """{"ms", "scores_per_s", "kernel", "frac", "parity_ok"} of one secondary block (everything else: bench_detail.json)"""
    if not isinstance(b, dict):
        return None
    if "error" in b and "ms_per_step" not in b:
        return {"error": str(b["error"])[:120]}
    o = {"ms": b.get("ms_per_step"), "scores_per_s": b.get("scores_per_s")}
    k = (b.get("kernels") or {}).get("crossprod")
    if isinstance(k, dict):
        o.update({"kernel": str(k.get("kernel"))[:40], "kernel_ms": k.get("kernel_ms"), "frac": k.get("frac")})
        if isinstance(k.get("lds_roof"), dict):
            o["lds_frac"] = k["lds_roof"].get("frac")
        for kk in ("lds_active", "clock_ghz"):
            if kk in k:
                o[kk] = k[kk]
    if isinstance(b.get("pipeline"), dict):
        o["pipeline_frac"] = b["pipeline"].get("frac")
    if "phases_ms" in b:
        o["phases_ms"] = {kk[:24]: round(v, 3) for kk, v in b["phases_ms"].items()}
    if isinstance(b.get("cpu_baseline"), dict):
        o["cpu_1core"] = b["cpu_baseline"].get("value")
    if "vs_baseline" in b:
        o["vs_published"] = b["vs_baseline"]
    if isinstance(b.get("mfma_backend"), dict) and "frac" in b["mfma_backend"]:
        o["mfma"] = _pick(b["mfma_backend"], ("ms", "achieved", "frac", "useful_frac"))
    o["parity_ok"] = _parity_ok(b.get("parity"))
    return o


def compact_line(full, detail_path="bench_detail.json"):
    """The ONE stdout line: the contract's keys, `roofline` (HBM `frac` and the LDS roof `lds_frac` side by side),
    `cpu_baseline`, one small summary object per secondary block.  Never longer than LINE_LIMIT bytes: block summaries are
    dropped (largest first) before anything of the contract is.  The complete record is `full` (bench_detail.json)."""
    if full.get("profile_only"):
        line = {"profile_only": True, "config": full.get("config"), "detail": detail_path}
        return json.dumps(line, separators=(",", ":"))
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "preheat_steps", "ms_per_step",
                        "ms_per_step_cold", "value_cold", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = _pick(full.get("config", {}), ("workload", "genes", "samples_per_gpu", "sets", "memberships", "parallelism"))
    r = full.get("roofline") or {}
    roof = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel_ms",
                     "traffic_source"))
    if isinstance(r.get("lds_roof"), dict):      # the roof the exact-fp64 LDS gather sits on, beside the HBM one
        roof["lds_frac"] = r["lds_roof"].get("frac")
        roof["lds_achieved"] = r["lds_roof"].get("achieved")
        roof["lds_peak"] = r["lds_roof"].get("peak")
    for kk in ("lds_active", "clock_ghz"):       # PMC of the same kernel and shape (profiles/traffic.json): LDS busy share, clock held
        if kk in r:
            roof[kk] = r[kk]
    if isinstance(full.get("pipeline"), dict):
        line["pipeline_frac"] = full["pipeline"].get("frac")
    line["roofline"] = roof
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        cb = _pick(c, ("value", "unit", "cores", "kind"))
        cb["sample"] = str(c.get("sample", ""))[:200]
        if isinstance(c.get("all_cores"), dict):
            cb["all_cores_value"], cb["all_cores"] = c["all_cores"].get("value"), c["all_cores"].get("cores")
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    if "phases_ms" in full:
        line["phases_ms"] = full["phases_ms"]
    line["parity_ok"] = _parity_ok(full.get("parity"))
    if isinstance(full.get("host_entry"), dict):
        line["host_entry_ms"] = full["host_entry"].get("ms")          # PCIe-inclusive, never `value`
    if isinstance(full.get("gather"), dict):
        line["gather"] = {k: _pick(v, ("completed", "ms", "GB/s")) for k, v in full["gather"].items() if isinstance(v, dict)}
    blocks = {}
    for name in ("c3", "c4", "c5_shard", "c3_real"):
        if name in full:
            blocks[name] = _block_summary(full[name])
    for name, b in (full.get("ref_shape") or {}).items() if isinstance(full.get("ref_shape"), dict) else ():
        blocks["ref_" + name] = _block_summary(b)
    line["blocks"] = blocks
    line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    while len(text.encode()) >= LINE_LIMIT and line["blocks"]:
        worst = max(line["blocks"], key=lambda k: len(json.dumps(line["blocks"][k])))
        for k in ("phases_ms", "mfma", "kernel"):                      # thin it first, drop it only then
            if isinstance(line["blocks"][worst], dict) and k in line["blocks"][worst]:
                del line["blocks"][worst][k]
                break
        else:
            del line["blocks"][worst]
        text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) >= LINE_LIMIT:                               # cannot happen with the keys above; never print it
        line["config"]["workload"] = line["config"].get("workload", "")[:120]
        line["cpu_baseline"] = _pick(line["cpu_baseline"] or {}, ("value", "unit", "cores", "kind"))
        text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) < LINE_LIMIT, len(text)
    return text


def emit(full, detail_path=None):
    """full record -> bench_detail.json (stderr too with BENCH_DETAIL_STDERR=1), compact line -> the LAST line of stdout"""
    detail_path = detail_path or os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    shown = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    try:
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1)
            fh.write("\n")
    except OSError as exc:
        shown = f"(not written: {exc})"
    if os.environ.get("BENCH_DETAIL_STDERR") == "1":      # opt-in: a box whose files do not come back
        print("[bench detail] " + json.dumps(full), file=sys.stderr, flush=True)
    else:
        print(f"[bench] full record: {shown}", file=sys.stderr, flush=True)
    try:    # whatever native libraries (RCCL's version banner) still hold in the C stdio buffer goes out first
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(compact_line(full, shown), flush=True)


# ----------------------------------------------------------------------------------------- N > 1: start the ranks
def launcher_argv(gpus, argv, port=None):
    """the command `bench.py --gpus N` starts when nothing launched it as a rank (the driver's own form, SCALE runs)"""
    port = port or os.environ.get("MASTER_PORT") or str(29500 + (os.getpid() % 400))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(a, argv):
    """Start N ranks as a CHILD process (never exec: this may run under a profiler that already initialised the GPU) and
    hand back its exit status; the child's rank 0 prints the line, which is forwarded untouched."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, _cpu_threads() // max(1, a.gpus))))
    cmd = launcher_argv(a.gpus, argv)
    print("[bench] starting ranks: " + " ".join(cmd), file=sys.stderr, flush=True)
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True,
                              timeout=float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "3000")))
    except subprocess.TimeoutExpired as exc:     # a hung rank: say so instead of waiting for the driver's own clock
        print(f"[bench] the ranks did not finish within {exc.timeout:.0f} s (BENCH_LAUNCH_TIMEOUT)", file=sys.stderr)
        return 4
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode == 0:
        try:
            got = json.loads(lines[-1]).get("n_gpus") if lines else None
        except ValueError:
            got = None
        if not lines or got is None:              # exit status 0 and no line would read as a successful, empty run
            print("[bench] the ranks exited with status 0 but printed no JSON line", file=sys.stderr)
            return 3
        if got != a.gpus:
            print(f"[bench] the ranks reported n_gpus = {got}, --gpus asked for {a.gpus}", file=sys.stderr)
            return 3
    return proc.returncode


def dry_run(a):
    """--dry-run: no GPU, no kernels.  Every rank joins the process group (gloo), the max-over-ranks reduction of a made-up
    elapsed time runs, rank 0 prints a line through the same `emit` -- so the launcher, the rendezvous and the line format can
    be tested on a CPU-only box.  The line says data = "dry-run" and carries no measurement."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if os.environ.get("BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
        raise SystemExit(f"bench.py --dry-run: rank {rank} fails on request (BENCH_DRY_FAIL_RANK)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, seen = float(t.item()), dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        elapsed, seen = 1.0, 1
    if rank == 0:
        g, n, m = a.genes, a.samples, a.sets
        emit({"metric": METRIC, "value": 0.0, "unit": "scores/s", "n_gpus": seen, "steps": a.steps, "warmup": a.warmup,
              "preheat_steps": a.preheat_steps, "ms_per_step": 1e3 * elapsed / max(1, a.steps), "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "dry-run (no GPU work, no measurement)",
              "config": {"workload": f"DRY RUN of C2 dense plaid(): {g} x {n}/GPU x {m}", "genes": g, "samples_per_gpu": n,
                         "sets": m, "parallelism": f"sample-shard x{seen}"},
              "roofline": None, "cpu_baseline": None})
    return 0


def _traffic(kernel, shape, columns=None):
    """(HBM bytes per launch, where they were measured) from the committed PMC passes (profiles/traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes -- counters cannot be
    collected inside the bench): per launch for the headline shape, per column x the columns of this launch for the
    kernels measured on a smaller panel of the same shape"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(path)).get(f"{kernel}/{shape}", {})
        src = e.get("source", "profiles/traffic.json")
        if columns is not None and "hbm_bytes_per_column" in e:
            return int(e["hbm_bytes_per_column"] * columns), src + f" (per column, measured on {e.get('measured_columns', '?')} columns)"
        v = e.get("hbm_bytes_per_launch")
        return (v, src) if v is not None else None
    except Exception:
        return None


def _pmc_extra(kernel, shape):
    """{lds_active, clock_ghz, lds_conflict_share, wait_share} of a kernel from the committed PMC passes (profiles/traffic.json:
    SQ_LDS_IDX_ACTIVE / CUs over GRBM_GUI_ACTIVE / 8, GRBM_GUI_ACTIVE / 8 / the launch's duration in that very pass, ...):
    what puts "it sits on the LDS roof" -- or does not -- into the record next to lds_frac"""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{kernel}/{shape}", {})
        o = {k: e[k] for k in ("lds_active", "clock_ghz", "lds_conflict_share", "wait_share") if k in e}
        if o:
            o["pmc_source"] = e.get("source", "profiles/traffic.json")
        return o
    except Exception:
        return {}


def _pipeline(kernels, ms_per_step, skip=()):
    """the WHOLE step against the HBM roof: the algorithmic bytes of every kernel that ran (SURVEY.md 8(d) per phase) over the
    step's time and 8 TB/s -- the view the per-kernel fractions do not give (a 14 ms shift at 0.7 of the roof beside a 20 ms
    crossprod at 0.25)"""
    tot = sum(int(k.get("algorithmic_bytes", 0)) for nm, k in kernels.items() if isinstance(k, dict) and nm not in skip)
    return {"algorithmic_bytes": tot, "frac": round(tot / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "sum of the phases' algorithmic bytes / ms_per_step / 8 TB/s"
                    + (f" (without {', '.join(skip)}: the fused crossprod made that sweep unnecessary)" if skip else "")}


def _roof(kernel, alg_bytes, ms, traffic=None, lds_bytes=None, extra=None):
    traffic_source = None
    if isinstance(traffic, tuple):
        traffic, traffic_source = traffic
    ach = alg_bytes / (ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "algorithmic_bytes": int(alg_bytes),
         "kernel_ms": round(ms, 4)}
    if traffic is not None:
        r["traffic_source"] = traffic_source or "profiles/traffic.json"   # (PMC passes cannot run inside the bench)
    if lds_bytes is not None:
        la = lds_bytes / (ms * 1e-3) / 1e9
        r["lds_roof"] = {"achieved": round(la, 1), "peak": LDS_PEAK_GBPS, "unit": "GB/s", "frac": round(la / LDS_PEAK_GBPS, 4),
                         "bytes": int(lds_bytes), "note": "bytes the kernel gathers from LDS per launch / kernel time"}
    if extra:
        r.update(extra)
    return r


def _fp64_roof(r, flop):
    """FP64-ALU roof of a gather kernel: SURVEY.md 8(d) counts 2 flop per (membership, sample); the kernel issues one fp64
    add for it, i.e. half a vector lane-slot of the FMA peak"""
    tf = flop / (r["kernel_ms"] * 1e-3) / 1e12
    r["fp64_alu_roof"] = {"achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                          "flop": flop}
    return r


class Events:
    """HIP events on the context's stream around the phases of a step (torch events see exactly that stream)"""

    def __init__(self, torch, stream, steps, marks):
        self.torch, self.stream, self.marks = torch, stream, marks
        self.ev = [[torch.cuda.Event(enable_timing=True) for _ in range(marks)] for _ in range(steps)]

    def rec(self, k, i):
        if k is not None:
            self.ev[k][i].record(self.stream)

    def phase_ms(self, i):
        import numpy as np
        return float(np.mean([e[i].elapsed_time(e[i + 1]) for e in self.ev]))


def _time_k(torch, dist, use_dist, dev, steps, step):
    """exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks"""
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def _timed(torch, dist, use_dist, dev, steps, warmup, step, preheat=0, cold_out=None):
    # cold_out (a dict): FIRST the contract's sequence without any pre-heat -- W warm-up steps, K timed steps -- is measured
    # and stored as cold_out["elapsed"], so that every line carries both numbers (rounds and the driver's SCALE runs stay
    # comparable whatever --preheat-steps they pass); then the pre-heated measurement below, which is the headline.
    if cold_out is not None and preheat > 0:
        for _ in range(warmup):
            step(None)
        cold_out["elapsed"] = _time_k(torch, dist, use_dist, dev, steps, lambda k: step(None))
    # `preheat` untimed steps BEFORE the W warmup steps of the contract: the part needs ~25 ms of continuous load to reach
    # its sustained clocks, and W = 5 steps of 1.2 ms are not that (measured on one box, K = 20: W = 5 1.19 ms per step,
    # W = 20 1.126, W = 50..1000 1.114..1.123).  The timed region is untouched: exactly K full steps between the two
    # barriers.  The count is fixed (not a wall time), so every rank runs the same collectives; it is reported in the line.
    for _ in range(preheat):
        step(None)
    if preheat:
        torch.cuda.synchronize()
    for _ in range(warmup):
        step(None)
    elapsed = _time_k(torch, dist, use_dist, dev, steps, step)
    if cold_out is not None and preheat <= 0:
        cold_out["elapsed"] = elapsed
    return elapsed


def _cpu_threads():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _rel_err(a, b):
    import numpy as np
    with np.errstate(all="ignore"):
        return float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-9)))



# ----------------------------------------------------------------------------------------- parity of the TIMED launches
def _snapshot(torch, S, med, red, flags, n, m, more=None):
    """probe columns (first / around element offset 2^31 / last, oracle.fullsize.probe_columns) of the result the LAST
    TIMED step left in S, plus the device's per-column vectors and scalars, copied to the host before anything else
    touches the buffers"""
    from oracle import fullsize
    cols, crosses = fullsize.probe_columns(n, m)
    idx = torch.as_tensor(cols, device=S.device)
    torch.cuda.synchronize()
    snap = {"cols": cols, "idx": idx, "crosses": crosses, "S": np_f(S.index_select(0, idx)),
            "med": med.cpu().numpy().copy(), "red": red.cpu().numpy().copy(), "flags": flags.cpu().numpy().copy()}
    for k, v in (more or {}).items():
        snap[k] = v.cpu().numpy().copy()
    return snap


def np_f(t):
    """(k, m) row-major device tensor -> (m, k) Fortran-ordered host array (R layout)"""
    import numpy as np
    return np.asfortranarray(t.cpu().numpy().T)


def _parity_report(res, snap, n, m, extra=None):
    from oracle import fullsize
    out = {"launch": "full", "checked": "the S the last timed step produced",
           "columns": int(len(snap["cols"])),
           "column_runs": [list(r) for r in fullsize.contiguous_runs(snap["cols"])],
           "offsets_past_2^31_checked": bool(snap["crosses"]),
           "last_element_offset": int(m) * int(n) - 1}
    out.update({k: (float(v) if isinstance(v, float) else v) for k, v in res.items()})
    if extra:
        out.update(extra)
    return out


def _probe_csc(fullsize, Xp, Xi, Xx, cols, more=None):
    """the CSC slots of the probe columns `cols` of a device-resident dgCMatrix, copied off the device run by run; `more`: a
    device vector parallel to Xx (e.g. the ranks of the stored values) -> its probe entries as a fourth result"""
    import numpy as np
    ph_all = Xp.cpu().numpy()
    parts, extra = [], []
    for lo, hi in fullsize.contiguous_runs(cols):
        q0, q1 = int(ph_all[lo]), int(ph_all[hi])
        parts.append((ph_all[lo:hi + 1], Xi[q0:q1].cpu().numpy(), Xx[q0:q1].cpu().numpy()))
        if more is not None:
            extra.append(more[q0:q1].cpu().numpy())
    sp_, si_, sx_ = fullsize.sub_csc(parts)
    return (sp_, si_, sx_) if more is None else (sp_, si_, sx_, np.concatenate(extra))


def _check_plaid_launch(torch, stream, snap, raw_oracle, relaunch_raw, S, n, m):
    """parity of a timed plaid() launch (C2, the reference shapes): `snap` holds probe columns of the normalised S the last
    timed step left, `raw_oracle` the oracle's un-normalised scores of those columns; `relaunch_raw()` enqueues ONE more
    full-size crossprod into S without normalisation, for the independent check of the flag words and of the raw scores.
    Returns the parity report (ok = False + the assertion's text when the checker raised)."""
    import numpy as np
    from oracle import fullsize
    try:
        with torch.cuda.stream(stream):
            relaunch_raw()
        torch.cuda.synchronize()
        res = fullsize.check_normalised(raw_oracle, snap["S"], snap["med"], snap["cols"], snap["red"], snap["flags"],
                                        _full_minmax(torch, S))
        raw_g = np_f(S.index_select(0, snap["idx"]))
        res["raw_max_rel_err_vs_oracle"] = _rel_err(raw_g, raw_oracle)
        np.testing.assert_allclose(raw_g, raw_oracle, rtol=fullsize.RTOL, atol=fullsize.ATOL)
        return _parity_report(res, snap, n, m)
    except AssertionError as exc:
        return {"launch": "full", "ok": False, "error": str(exc)[:400]}


def _full_minmax(torch, Sraw):
    """independent device reduction over a FULL un-normalised result: (min, any zero) -- what R/plaid.R:556-557 tests"""
    return float(Sraw.min().item()), bool((Sraw == 0).any().item())



def _bench_gather(env, S, n_total, modes):
    """reassemble the score matrix on rank 0 (R/plaid.R:110-119 fills ONE matrix): timed separately, never part of
    `value`.  "host": every rank copies its block into its rows of one shared host matrix (completes at any size the
    host can hold); "device": slabs peer->root over xGMI (fp64 when it fits the root's free HBM, else fp32).  A gather
    that cannot complete is REFUSED before anything is allocated and reported as such -- never an exception."""
    torch, dist, rank = env["torch"], env["dist"], env["rank"]
    from plaid_amd import sharded
    world = env["world"]
    m = S.shape[1]
    out = {}
    for mode in modes:
        for dt in ((None,) if mode == "host" else (None, torch.float32)):
            key = mode if dt is None else f"{mode}_fp32"
            if key == "device_fp32" and out.get("device", {}).get("completed"):
                continue
            try:
                dist.barrier()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                full = sharded.gather_scores(S, n_total, dst=0, to=mode, dtype=dt)
                torch.cuda.synchronize()
                dist.barrier()
                tg = time.perf_counter() - tg
                item = 4 if dt is not None else 8
                moved = (n_total - (S.shape[0] if mode == "device" else 0)) * m * item
                out[key] = {"completed": True, "ms": round(1e3 * tg, 2), "GB": round(moved / 1e9, 2),
                            "GB/s": round(moved / tg / 1e9, 1), "dtype": "f32" if dt is not None else "f64"}
                del full
            except sharded.GatherRefused as exc:
                out[key] = {"completed": False, "refused": str(exc)[:300], "needed_GB": round(exc.needed / 1e9, 1),
                            "available_GB": round(exc.available / 1e9, 1)}
            torch.cuda.empty_cache()
    out["note"] = ("GB = bytes that crossed a link (device: the peers' blocks into the root over xGMI; host: every block over "
                   "its own PCIe link into one shared host matrix); not part of `value`")
    return out



def _fused_info(ctx, torch):
    """how the medians of the last timed step came about (status words of the fused crossprod, copied off the device)"""
    try:
        import ctypes as _C
        import numpy as _np
        nf, p_status, p_cal, _ = ctx.dev_fused_medians_info()
        if not nf:
            return None
        torch.cuda.synchronize()
        st_h, cal_h = _np.zeros(nf, dtype=_np.int32), _np.zeros(4)
        ctx.lib.plaidhip_memcpy_d2h(ctx.handle, st_h.ctypes.data_as(_C.c_void_p), _C.c_void_p(p_status), _C.c_size_t(4 * nf))
        ctx.lib.plaidhip_memcpy_d2h(ctx.handle, cal_h.ctypes.data_as(_C.c_void_p), _C.c_void_p(p_cal), _C.c_size_t(32))
        return {"columns": int(nf), "selected_from_candidates": int(st_h.sum()), "left_to_the_standalone_kernel": int(nf - st_h.sum()),
                "bracket": {"offset_from_predicted_mean": float(cal_h[0]), "half_width": float(cal_h[1]), "ignore_zero": bool(cal_h[2])},
                "note": "medians selected inside the crossprod launch (plaidhip_dev_spmm_*_fused_f64 + ..._resume): "
                        "phases_ms.crossprod includes the calibration, the column-mean prediction and the classifying "
                        "epilogue; phases_ms.col_medians+sum is the selection + the standalone kernel on the columns left"}
    except Exception as exc:  # pragma: no cover
        return {"error": str(exc)[:200]}


# ----------------------------------------------------------------------------------------- C2 (headline)
def run_c2(a, env):
    import numpy as np
    torch, dist, ctx, dev, stream = env["torch"], env["dist"], env["ctx"], env["dev"], env["stream"]
    world, rank, use_dist = env["world"], env["rank"], env["use_dist"]
    from plaid_amd import synth
    g, n, m = a.genes, a.samples, a.sets
    Gp, Gi = synth.geneset_csc(g, m)
    z = int(Gp[-1])
    gs = ctx.geneset(g, Gp, Gi)
    info = gs.info()
    X = torch.empty((n, g), dtype=torch.float64, device=dev)      # row-major (n, g) == R's g x n
    col0 = rank * n                                               # this rank's sample shard
    want_host = world == 1 and rank == 0 and a.cpu_sample > 0     # the host-entry leg needs the matrix in host memory
    Xhost = np.empty((g, n), dtype=np.float64, order="F") if want_host else None
    for j0 in range(0, n, 1024):
        j1 = min(n, j0 + 1024)
        blk = synth.dense_columns(g, col0 + j0, col0 + j1)        # (g, b) Fortran
        X[j0:j1].copy_(torch.from_numpy(np.ascontiguousarray(blk.T)))
        if want_host:
            Xhost[:, j0:j1] = blk
    S = torch.empty((n, m), dtype=torch.float64, device=dev)      # == m x n column-major
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    med = torch.empty(n, dtype=torch.float64, device=dev)
    red = torch.zeros(4, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ev = Events(torch, stream, a.steps, 4)

    def step(k=None):
        with torch.cuda.stream(stream):
            flags.zero_()
            ev.rec(k, 0)
            ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
            ev.rec(k, 1)
            if use_dist:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)                  # min(x)==0 over all shards
            ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
            ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
            ev.rec(k, 2)
            if use_dist:
                dist.all_reduce(red, op=dist.ReduceOp.SUM)                    # mean(medx) over all shards
            ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
            ev.rec(k, 3)

    cold = {}
    elapsed = _timed(torch, dist, use_dist, dev, a.steps, a.warmup, step, preheat=a.preheat_steps, cold_out=cold)
    snap = _snapshot(torch, S, med, red[0:2], flags, n, m) if (rank == 0 and world == 1 and a.cpu_sample > 0) else None
    ms_step = 1e3 * elapsed / a.steps
    spmm_ms, med_ms, shift_ms = ev.phase_ms(0), ev.phase_ms(1), ev.phase_ms(2)
    scores = float(world) * n * m
    value = scores / (elapsed / a.steps)

    # secondary, never in `value`: the opt-in mixed-precision crossprod (fp32 operand staging, fp64 sums)
    mixed = None
    if not a.no_mixed and world == 1:
        try:
            ctx.set_precision("mixed")
            msteps = max(3, min(a.steps, 10))
            step()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            for _ in range(msteps):
                step()
            torch.cuda.synchronize()
            tm = (time.perf_counter() - tm) / msteps
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

    # roofline of the dominant kernel: SURVEY.md 8(d): g*n*b + (4 z + 4 (m+1)) + m*n*b with b = 8
    alg_bytes = g * n * 8 + 4 * z + 4 * (m + 1) + m * n * 8
    spmm_kernel = "spmm_colpair_f64"   # (any leading dimension: the 16-byte loads need no 16-byte alignment)
    # every padded membership slot of the plan returns 8 bytes per sample column from LDS
    lds_bytes = float(info["padded_slots"]) * 8.0 * n
    roofline = _fp64_roof(_roof(spmm_kernel, alg_bytes, spmm_ms, _traffic(spmm_kernel, f"{g}x{n}x{m}"), lds_bytes,
                                extra=_pmc_extra(spmm_kernel, f"{g}x{n}x{m}")), 2.0 * z * n)
    kernels = {
        "col_medians": _roof("col_medians_wave_kernel" if m <= 6144 else "col_medians_stream_kernel", 8.0 * m * n, med_ms,
                             _traffic("col_medians_wave_kernel", f"{n}x{m}") if m <= 6144 else None),
        "shift_columns": _roof("shift_columns_kernel", 16.0 * m * n, shift_ms, _traffic("shift_columns_kernel", f"{n}x{m}")),
    }

    gather = _bench_gather(env, S, world * n, ("device",)) if (use_dist and not a.no_gather) else None

    cpu = None
    parity = None
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        from oracle import c_oracle
        nc = min(a.cpu_sample, n)
        Xh = synth.dense_columns(g, 0, nc)
        t1 = time.perf_counter()
        Sraw = c_oracle.plaid_dense(Xh, Gp, Gi, "mean", False)
        t2 = time.perf_counter()
        c_oracle.normalize_medians(Sraw.copy())
        t3 = time.perf_counter()
        nt = _cpu_threads()
        t4 = time.perf_counter()
        Sall = c_oracle.crossprod_dense(Xh, Gp, Gi, "mean", nt)
        c_oracle.normalize_medians_mt(Sall, None, nt)
        t5 = time.perf_counter()
        cpu = {"value": round(m * nc / (t3 - t1), 1), "unit": "scores/s", "cores": 1, "kind": "port",
               "sample": f"first {nc} of {n} sample columns x {m} sets, plain-C oracle (oracle/plaid_oracle.c): "
                         f"crossprod {t2 - t1:.2f} s + normalize_medians {t3 - t2:.2f} s",
               "all_cores": {"value": round(m * nc / (t5 - t4), 1), "cores": nt,
                             "note": "same code, OpenMP over sample columns (not what the single-threaded reference does)"},
               "cpu_count": os.cpu_count()}
        # parity of the launch that was timed: probe columns of ITS result against the oracle, column by column
        raw_o = c_oracle.crossprod_dense(np_f(X.index_select(0, snap["idx"])), Gp, Gi, "mean", nt)
        parity = _check_plaid_launch(torch, stream, snap, raw_o,
                                     lambda: ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, None), S, n, m)
        if mixed is not None and "error" not in mixed:
            ctx.set_precision("mixed")
            with torch.cuda.stream(stream):
                ctx.dev_spmm_dense(gs, X.data_ptr(), g, nc, S.data_ptr(), m, "mean", 1.0, 0.0, None)
            torch.cuda.synchronize()
            ctx.set_precision("f64")
            Sm = S[:nc].cpu().numpy().T
            mixed["max_rel_err_vs_oracle"] = float(np.max(np.abs(Sm - Sraw) / np.maximum(np.abs(Sraw), 1e-300)))
    # the host-pointer entry point an R session calls (PCIe-inclusive, never `value`): pageable host X in, S out
    host_entry = None
    if want_host:
        try:
            ctx.plaid_dense(Xhost[:, :256], Gp, Gi)                 # pins the staging buffers, caches the gene sets
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                Sh = ctx.plaid_dense(Xhost, Gp, Gi)
                ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            err = None
            if parity is not None:
                with torch.cuda.stream(stream):
                    step()
                torch.cuda.synchronize()
                err = float(np.max(np.abs(Sh[:, :64] - S[:64].cpu().numpy().T)))
            host_entry = {"entry": "plaidhip_plaid_dense", "ms": round(1e3 * min(ts), 2), "scores_per_s": round(m * n / min(ts), 1),
                          "GB/s_over_pcie": round((g * n * 8 + m * n * 8) / min(ts) / 1e9, 1),
                          "max_abs_diff_vs_device_path": err,
                          "note": "pageable host X (1.6 GB) -> pinned staging -> HBM, crossprod per landed column panel, "
                                  "normalize_medians, S (0.4 GB) back; includes everything an R caller waits for"}
            del Sh
        except Exception as exc:  # pragma: no cover
            host_entry = {"error": str(exc)[:200]}
    del Xhost
    out = {
        "value": value, "ms_per_step": ms_step, "ms_per_step_cold": 1e3 * cold["elapsed"] / a.steps,
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        "host_entry": host_entry,
        "gather": gather, "mixed_precision": mixed, "kernels": kernels, "pipeline": _pipeline(dict(kernels, crossprod=roofline), ms_step),
        "phases_ms": {"spmm": round(spmm_ms, 4), "col_medians+sum": round(med_ms, 4), "shift": round(shift_ms, 4),
                      "normalize_medians": round(med_ms + shift_ms, 4)},
        "config": {"workload": f"C2 dense plaid(): {g} genes x {n} samples/GPU x {m} gene sets "
                               f"(z={z} memberships), inputs resident in HBM",
                   "genes": g, "samples_per_gpu": n, "sets": m, "memberships": z,
                   "parallelism": f"sample-shard x{world}"},
    }
    gs.close()
    del X, S
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------- sparse X generator
def ssgsea_csc_oracle(c_oracle, np, Xp, Xi, Xx, g, Gp, Gi, alpha, threads):
    """replaid.ssgsea on a dgCMatrix, phase by phase, on the C oracle (R/plaid.R:244-255)"""
    t = [time.perf_counter()]
    r = c_oracle.sparse_colranks_mt(Xp, Xx, "average", False, threads)
    r = r ** (1.0 + alpha)
    gmax = max(r.max() if r.size else 0.0, 0.0)
    t.append(time.perf_counter())
    k = np.diff(Gp).astype(np.float64)
    S = c_oracle.crossprod_csc(Xp, Xi, r, g, Gp, Gi, "mean", threads)
    S /= gmax
    S -= (0.5 * k / (1e-8 + k))[:, None]                    # the "- 0.5" of :251 reaches every gene of a set
    t.append(time.perf_counter())
    c_oracle.normalize_medians_mt(S, None, threads)
    t.append(time.perf_counter())
    return S, [t[i + 1] - t[i] for i in range(3)]


def ssgsea_dense_oracle(c_oracle, np, X, Gp, Gi, alpha, threads):
    t = [time.perf_counter()]
    R = c_oracle.colranks_dense_mt(X, "average", False, threads)
    R **= (1.0 + alpha)
    R /= R.max()
    R -= 0.5
    t.append(time.perf_counter())
    S = c_oracle.crossprod_dense(R, Gp, Gi, "mean", threads)
    t.append(time.perf_counter())
    c_oracle.normalize_medians_mt(S, None, threads)
    t.append(time.perf_counter())
    return S, [t[i + 1] - t[i] for i in range(3)]


# ----------------------------------------------------------------------------------------- C3 / C5 (sparse ssGSEA)
def run_sparse_ssgsea(a, env, n, label, collective, real_sets=False):
    """`real_sets`: the collection has the shape of the reference's own (synth.geneset_csc_real: sizes 3 ... 5,000 + an
    all-genes set, hub genes, duplicated sets) instead of 15 ... 500 uniformly chosen genes; no CPU-baseline leg then"""
    import numpy as np
    torch, dist, ctx, dev, stream = env["torch"], env["dist"], env["ctx"], env["dev"], env["stream"]
    world, rank = env["world"], env["rank"]
    from plaid_amd import synth
    g, m, alpha = a.genes, a.big_sets, 0.25
    Gp, Gi = synth.geneset_csc_real(g, m) if real_sets else synth.geneset_csc(g, m)
    z = int(Gp[-1])
    t0 = time.perf_counter()
    gs = ctx.geneset(g, Gp, Gi)
    t_plan = time.perf_counter() - t0
    with torch.cuda.stream(stream):
        Xp, Xi, Xx, nnz, max_nnz = synth.device_sparse_cells(torch, dev, g, n, 20250615 + 7919 * rank)
        Rx = torch.empty_like(Xx)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        colmax = torch.zeros(n, dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        small = torch.zeros(8, dtype=torch.float64, device=dev)     # [0:2] {sum, count}; [2] max(rX)
    red, gmax = small[0:2], small[2:3]
    torch.cuda.synchronize()
    steps = a.block_steps
    ev = Events(torch, stream, steps, 5)

    def step(k=None):
        with torch.cuda.stream(stream):
            flags.zero_()
            ev.rec(k, 0)
            ctx.dev_colranks_csc(Xp.data_ptr(), Xx.data_ptr(), n, max_nnz, Rx.data_ptr(), "average", False, 1.0 + alpha,
                                 colmax.data_ptr())
            ctx.dev_max(colmax.data_ptr(), n, gmax.data_ptr())
            ev.rec(k, 1)
            if collective:
                dist.all_reduce(gmax, op=dist.ReduceOp.MAX)                   # max(rX) over all shards
            # the crossprod also classifies its scores for normalize_medians (plaidhip_dev_spmm_csc_fused_f64); the resume
            # call selects the medians among the candidates and sweeps only the columns it could not resolve
            ctx.dev_spmm_csc_fused(gs, Xp.data_ptr(), Xi.data_ptr(), Rx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, -0.5,
                                   flags.data_ptr(), None, gmax.data_ptr(), nnz=nnz)
            ev.rec(k, 2)
            if collective:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
            ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
            ev.rec(k, 3)
            if collective:
                dist.all_reduce(red, op=dist.ReduceOp.SUM)
            ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
            ev.rec(k, 4)

    elapsed = _timed(torch, dist, collective, dev, steps, 1, step)
    snap = None
    if not collective and rank == 0 and a.cpu_sample > 0:
        snap = _snapshot(torch, S, med, red, flags, n, m, {"colmax": colmax, "gmax": gmax})
    fused_info = _fused_info(ctx, torch)
    rank_ms, spmm_ms, med_ms, shift_ms = (ev.phase_ms(i) for i in range(4))
    scatter = nnz * 8 < g * n
    spmm_alg = 12.0 * nnz + 4.0 * (n + 1) + 4.0 * z + 4.0 * (m + 1) + 8.0 * m * n
    out = {
        "workload": f"{label}: replaid.ssgsea(alpha={alpha}) on sparse {g} genes x {n} cells"
                    f"{'/GPU' if collective else ''} ({nnz} stored values, {100.0 * nnz / (g * n):.2f} %) x {m} gene sets "
                    f"{'of realistic shape (sizes 3..' + str(g) + ', Zipf gene popularity, an all-genes set, duplicated sets) ' if real_sets else ''}"
                    f"(z={z}), fp64, inputs resident in HBM",
        "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
        "scores_per_s": round(float(world if collective else 1) * n * m / (elapsed / steps), 1),
        "phases_ms": {"sparse_colranks+max": round(rank_ms, 4), "crossprod": round(spmm_ms, 4),
                      "col_medians+sum": round(med_ms, 4), "shift": round(shift_ms, 4)},
        "rank_keys_per_s": round(nnz / (rank_ms * 1e-3), 1),
        "geneset_plan_s": round(t_plan, 2),
        "fused_medians": fused_info,
        "kernels": {
            "sparse_colranks": _roof("colranks_bucket_kernel<256,8>" if max_nnz <= 2048 else "colranks_bucket_kernel", 16.0 * nnz, rank_ms,
                                     _traffic("colranks_bucket_kernel<256,8>", "csc", n) if max_nnz <= 2048 else None),
            "crossprod": _roof("spmm_scatter_csc_f64" if scatter else "spmm_colpair_f64<csc>", spmm_alg, spmm_ms,
                               _traffic("spmm_scatter_csc_f64<med>" if (fused_info and fused_info.get("columns")) else "spmm_scatter_csc_f64", f"{g}xNx{m}", n) if scatter else None,
                               extra=dict({"lds_atomic_adds_per_s": round(nnz * (z / g) / (spmm_ms * 1e-3), 1)},
                                          **_pmc_extra("spmm_scatter_csc_f64<med>" if (fused_info and fused_info.get("columns")) else "spmm_scatter_csc_f64", f"{g}xNx{m}")) if scatter else None),
            "col_medians": _roof("median_select_kernel + col_medians_stream_kernel (unresolved columns)", 8.0 * m * n, med_ms, None,
                                 extra={"note": "algorithmic bytes are those of the full sweep the fused crossprod made unnecessary: "
                                                "frac > 1 would only say that S was not read again"}),
            "shift_columns": _roof("shift_columns_kernel", 16.0 * m * n, shift_ms, _traffic("shift_columns_kernel", f"Nx{m}", n)),
        },
    }
    out["pipeline"] = _pipeline(out["kernels"], out["ms_per_step"], skip=("col_medians",) if (fused_info and fused_info.get("columns")) else ())
    if collective and not a.no_gather:
        # (the host gather of config 5 writes 8 x 50 GB through /dev/shm: tens of seconds; it runs on request only, so that
        # the driver's scaling runs stay within minutes -- measured in tools/gather_host_check.py and DESIGN.md 8)
        out["gather"] = _bench_gather(env, S, world * n, ("host", "device") if a.host_gather else ("device",))
    if not collective and rank == 0 and a.cpu_sample > 0:
        from oracle import c_oracle
        nc = min(2048, n)
        ph = Xp[:nc + 1].cpu().numpy()
        zc = int(ph[-1])
        ih, xh = Xi[:zc].cpu().numpy(), Xx[:zc].cpu().numpy()
        if not real_sets:
            S1, t1 = ssgsea_csc_oracle(c_oracle, np, ph, ih, xh, g, Gp, Gi, alpha, 1)
            nt = _cpu_threads()
            _, tn = ssgsea_csc_oracle(c_oracle, np, ph, ih, xh, g, Gp, Gi, alpha, nt)
            out["cpu_baseline"] = {
                "value": round(m * nc / sum(t1), 1), "unit": "scores/s", "cores": 1, "kind": "port",
                "sample": f"first {nc} of {n} cells x {m} sets, plain-C oracle: sparse_colranks+pow {t1[0]:.2f} s, crossprod "
                          f"(Gustavson order) {t1[1]:.2f} s, normalize_medians {t1[2]:.2f} s",
                "all_cores": {"value": round(m * nc / sum(tn), 1), "cores": nt}, "cpu_count": os.cpu_count()}
        # checker: probe cells of the launch that was timed (first / around element offset 2^31 / last), phase by phase
        from oracle import fullsize
        try:
            cols = snap["cols"]
            sp_, si_, sx_, rx_gpu = _probe_csc(fullsize, Xp, Xi, Xx, cols, more=Rx)
            gmax_dev = float(snap["gmax"][0])
            assert gmax_dev == float(snap["colmax"].max()), "max(rX) on the device != max of the device's colmax[]"
            r_o, raw_o = fullsize.ssgsea_csc_raw(sp_, si_, sx_, g, Gp, Gi, alpha, gmax_dev)
            ranks_exact = bool(np.array_equal(fullsize.ranks_from_powered(rx_gpu, 1.0 + alpha), r_o))
            assert ranks_exact, "sparse_colranks of the probe cells differ from the oracle"
            cm = np.array([rx_gpu[sp_[j]:sp_[j + 1]].max() if sp_[j + 1] > sp_[j] else 0.0 for j in range(len(cols))])
            assert np.array_equal(cm, snap["colmax"][cols]), "colmax[] of the probe cells != max of their powered ranks"
            with torch.cuda.stream(stream):     # one more full-size launch, un-normalised: for the independent flag check
                S2 = torch.empty_like(S)
                ctx.dev_spmm_csc_ranks(gs, Xp.data_ptr(), Xi.data_ptr(), Rx.data_ptr(), n, S2.data_ptr(), m, gmax.data_ptr(),
                                       "mean", 1.0, -0.5, None, nnz=nnz)
            torch.cuda.synchronize()
            res = fullsize.check_normalised(raw_o, snap["S"], snap["med"], cols, snap["red"], snap["flags"],
                                            _full_minmax(torch, S2))
            del S2
            res["ranks_bit_exact"] = ranks_exact
            res["gmax_equals_max_of_colmax"] = True
            out["parity"] = _parity_report(res, snap, n, m, {"note": "scores are centred (|s| <~ 0.5): abs error is the meaningful one"})
        except AssertionError as exc:
            out["parity"] = {"launch": "full", "ok": False, "error": str(exc)[:400]}
    gs.close()
    del Xp, Xi, Xx, Rx, S
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------- C4 (dense ssGSEA)
def run_c4(a, env):
    import numpy as np
    torch, dist, ctx, dev, stream = env["torch"], env["dist"], env["ctx"], env["dev"], env["stream"]
    from plaid_amd import synth
    g, n, m, alpha = a.genes, a.c4_samples, a.big_sets, 0.25
    Gp, Gi = synth.geneset_csc(g, m)
    z = int(Gp[-1])
    gs = ctx.geneset(g, Gp, Gi)
    info = gs.info()
    gen = torch.Generator(device=dev)
    gen.manual_seed(20250614)
    with torch.cuda.stream(stream):
        X = torch.empty((n, g), dtype=torch.float64, device=dev)
        for j0 in range(0, n, 8192):
            j1 = min(n, j0 + 8192)
            X[j0:j1] = torch.randn((j1 - j0, g), dtype=torch.float64, device=dev, generator=gen) * 2.0 + 8.0   # N(8, 2^2)
        R = torch.empty_like(X)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        colmax = torch.zeros(n, dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        small = torch.zeros(8, dtype=torch.float64, device=dev)
    red, gmax = small[0:2], small[2:3]
    torch.cuda.synchronize()
    steps = a.block_steps
    ev = Events(torch, stream, steps, 5)

    def pipeline(ncols, k=None):
        flags.zero_()
        ev.rec(k, 0)
        ctx.dev_colranks_dense(X.data_ptr(), g, g, ncols, R.data_ptr(), g, "average", False, 1.0 + alpha, colmax.data_ptr())
        ctx.dev_max(colmax.data_ptr(), ncols, gmax.data_ptr())
        ev.rec(k, 1)
        # the crossprod also classifies its scores for normalize_medians (plaidhip_dev_spmm_dense_fused_f64, round 5); the resume
        # call selects the medians among the candidates and sweeps only the columns it could not resolve
        ctx.dev_spmm_dense_fused(gs, R.data_ptr(), g, ncols, S.data_ptr(), m, "mean", 1.0, -0.5, flags.data_ptr(), gmax.data_ptr())
        ev.rec(k, 2)
        ctx.dev_col_medians_resume(S.data_ptr(), m, m, ncols, None, med.data_ptr(), flags.data_ptr())
        ctx.dev_sum(med.data_ptr(), ncols, red.data_ptr())
        ev.rec(k, 3)
        ctx.dev_shift_columns(S.data_ptr(), m, m, ncols, med.data_ptr(), 0.0, red.data_ptr())
        ev.rec(k, 4)

    def step(k=None):
        with torch.cuda.stream(stream):
            pipeline(n, k)

    elapsed = _timed(torch, dist, False, dev, steps, 1, step)
    fused_info = _fused_info(ctx, torch)
    snap = None
    if a.cpu_sample > 0:
        snap = _snapshot(torch, S, med, red, flags, n, m, {"colmax": colmax, "gmax": gmax})
        snap["R"] = R.index_select(0, snap["idx"]).cpu().numpy()          # (k, g): powered ranks of the probe samples
    rank_ms, spmm_ms, med_ms, shift_ms = (ev.phase_ms(i) for i in range(4))
    spmm_alg = 8.0 * g * n + 4.0 * z + 4.0 * (m + 1) + 8.0 * m * n
    out = {
        "workload": f"C4: replaid.ssgsea(alpha={alpha}) on dense {g} genes x {n} samples x {m} gene sets (z={z}), fp64, "
                    "crossprod with the sparse membership (what the reference multiplies, R/plaid.R:253 -> :80), "
                    "inputs resident in HBM",
        "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
        "scores_per_s": round(float(n) * m / (elapsed / steps), 1),
        "phases_ms": {"colranks+pow+max": round(rank_ms, 4), "crossprod": round(spmm_ms, 4),
                      "col_medians+sum": round(med_ms, 4), "shift": round(shift_ms, 4)},
        "rank_keys_per_s": round(float(g) * n / (rank_ms * 1e-3), 1),
        "fused_medians": fused_info,
        "kernels": {
            "colranks": _roof("colranks_bucket_kernel<1024,20>", 16.0 * g * n, rank_ms,
                              _traffic("colranks_bucket_kernel<1024,20>", f"{g}xN", n)),
            # (with the medians selected in the launch the timed phase is the fused call: the 256-column calibration launch,
            #  its standalone medians, the calibration and the classifying kernel -- labelled and priced as that kernel)
            "crossprod": _fp64_roof(_roof("spmm_colpair_f64<med> (+ 256-column calibration)" if fused_info else "spmm_colpair_f64",
                                          spmm_alg, spmm_ms,
                                          _traffic("spmm_colpair_f64<med>" if fused_info else "spmm_colpair_f64", f"{g}xNx{m}", n),
                                          lds_bytes=float(info["padded_slots"]) * 8.0 * n,
                                          extra=_pmc_extra("spmm_colpair_f64<med>" if fused_info else "spmm_colpair_f64", f"{g}xNx{m}")), 2.0 * z * n),
            "col_medians": (_roof("median_select_kernel + col_medians_stream_kernel (unresolved columns)", 8.0 * m * n, med_ms, None,
                                  extra={"note": "algorithmic bytes are those of the full sweep the fused crossprod made unnecessary: "
                                                 "frac > 1 would only say that S was not read again"})
                            if fused_info else
                            _roof("col_medians_stream_kernel", 8.0 * m * n, med_ms, _traffic("col_medians_stream_kernel", f"Nx{m}", n))),
            "shift_columns": _roof("shift_columns_kernel", 16.0 * m * n, shift_ms, _traffic("shift_columns_kernel", f"Nx{m}", n)),
        },
        "dense_gemm_equivalent": {"flop": 2.0 * g * n * m, "note": "the dense contraction config 4 names would be "
                                  f"{2.0 * g * n * m:.2e} flop (x3 issues as a bf16 split) against {2.0 * z * n:.2e} for this "
                                  "crossprod: see DESIGN.md (C4 row) for the measured MFMA rate beside this time"},
    }
    out["pipeline"] = _pipeline(out["kernels"], out["ms_per_step"], skip=("col_medians",) if fused_info else ())
    # the same crossprod as the dense contraction config 4 names, on the matrix cores (opt-in backend): ONE call at the
    # FULL size (all n samples x m sets; the backend walks 8,192-sample panels), timed with HIP events, next to the SpMM
    # kernel's time for the same launch (phases_ms.crossprod above).  Its probe columns are checked against the oracle in
    # the parity leg below.  R still holds the powered average ranks of the timed steps.
    mfma_probe = None
    try:
        ctx.set_option("spmm_dense_kernel", "mfma")
        with torch.cuda.stream(stream):   # builds the dense bf16 G (2 GB) and sizes the workspace: not part of the timing
            ctx.dev_spmm_dense(gs, R.data_ptr(), g, min(n, 256), S.data_ptr(), m, "mean", 1.0, -0.5, None, gmax.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            ctx.dev_spmm_dense(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, -0.5, None, gmax.data_ptr())
            e1.record(stream)
        torch.cuda.synchronize()
        t_mfma = e0.elapsed_time(e1)
        if snap is not None:
            mfma_probe = np_f(S.index_select(0, snap["idx"]))
        flop = 3.0 * 2.0 * g * float(n) * m
        tf = flop / (t_mfma * 1e-3) / 1e12
        out["mfma_backend"] = {
            "kernel": "crossprod_mfma_bf16x3_kernel", "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0,
            "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "useful_frac": round(2.0 * g * float(n) * m / (t_mfma * 1e-3) / 1e12 / 2500.0, 4),   # the contraction's own 2 g n m flop: `frac` counts the bf16 x 3 split's three issues
            "samples": n, "sets": m, "genes": g, "launches": 1,
            "ms": round(t_mfma, 3), "spmm_ms_same_launch": round(spmm_ms, 3), "slowdown_vs_spmm": round(t_mfma / spmm_ms, 2),
            "flop": flop, "scores_per_s": round(float(n) * m / (t_mfma * 1e-3), 1),
            "note": "dense 0/1 G (bf16, 2 GB) x bf16x3 split of the rank weights, fp32 accumulate: 3 x 2 g n m flop against "
                    "2 z n for the SpMM (z/(g m) = 0.7 % dense); round 5: 256 x 256 tiles, 512 threads, two LDS stages; the time "
                    "includes the split of every 8,192-sample panel into three bf16 planes; opt-in backend, never the default"}
    except Exception as exc:  # pragma: no cover
        out["mfma_backend"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
    finally:
        ctx.set_option("spmm_dense_kernel", "auto")
    if a.cpu_sample > 0:
        from oracle import c_oracle
        nc = min(512, n)
        Xh = np.asfortranarray(X[:nc].cpu().numpy().T)
        S1, t1 = ssgsea_dense_oracle(c_oracle, np, Xh.copy(order="F"), Gp, Gi, alpha, 1)
        nt = _cpu_threads()
        _, tn = ssgsea_dense_oracle(c_oracle, np, Xh.copy(order="F"), Gp, Gi, alpha, nt)
        out["cpu_baseline"] = {
            "value": round(m * nc / sum(t1), 1), "unit": "scores/s", "cores": 1, "kind": "port",
            "sample": f"first {nc} of {n} samples x {m} sets, plain-C oracle: colranks+pow {t1[0]:.2f} s, crossprod "
                      f"{t1[1]:.2f} s, normalize_medians {t1[2]:.2f} s",
            "all_cores": {"value": round(m * nc / sum(tn), 1), "cores": nt}, "cpu_count": os.cpu_count()}
        # checker: probe samples of the launch that was timed (first / around element offset 2^31 / last), phase by phase
        from oracle import fullsize
        try:
            cols = snap["cols"]
            gmax_dev = float(snap["gmax"][0])
            assert gmax_dev == float(snap["colmax"].max()), "max(rX) on the device != max of the device's colmax[]"
            Xc = np_f(X.index_select(0, snap["idx"]))
            r_o, raw_o = fullsize.ssgsea_dense_raw(Xc, Gp, Gi, alpha, gmax_dev)
            ranks_exact = bool(np.array_equal(fullsize.ranks_from_powered(snap["R"].T, 1.0 + alpha), r_o))
            assert ranks_exact, "colranks of the probe samples differ from the oracle"
            assert np.array_equal(snap["R"].max(axis=1), snap["colmax"][cols]), "colmax[] != max of the powered ranks"
            with torch.cuda.stream(stream):     # one more full-size launch, un-normalised: for the independent flag check
                ctx.dev_spmm_dense(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, -0.5, None, gmax.data_ptr())
            torch.cuda.synchronize()
            res = fullsize.check_normalised(raw_o, snap["S"], snap["med"], cols, snap["red"], snap["flags"],
                                            _full_minmax(torch, S))
            res["ranks_bit_exact"] = ranks_exact
            res["gmax_equals_max_of_colmax"] = True
            out["parity"] = _parity_report(res, snap, n, m, {"note": "scores are centred (|s| <~ 0.5): abs error is the meaningful one"})
            if mfma_probe is not None and "error" not in out.get("mfma_backend", {}):
                # the MFMA backend's un-normalised scores of the same probe samples against the same oracle columns
                err = float(np.max(np.abs(mfma_probe - raw_o)))
                ok = bool(np.allclose(mfma_probe, raw_o, rtol=fullsize.RTOL, atol=1e-6))
                out["mfma_backend"]["parity"] = {
                    "launch": "full", "ok": ok, "columns": int(len(cols)), "offsets_past_2^31_checked": bool(snap["crosses"]),
                    "max_abs_err_vs_oracle": err, "tolerance": "rtol 1e-5 + atol 1e-6 (bf16x3 split: 24 significant bits, fp32 sums)",
                    "max_abs_diff_vs_spmm": float(np.max(np.abs(mfma_probe - np_f(S.index_select(0, snap["idx"])))))}
        except AssertionError as exc:
            out["parity"] = {"launch": "full", "ok": False, "error": str(exc)[:400]}
    # (behind the parity leg: it overwrites R with the min-ties ranks)
    # replaid.sing (R/plaid.R:213-219) at the same size: min-ties ranks, then the crossprod of the RANK matrix under the
    # three exact stagings (u16: four samples per LDS entry, integer sums -- the default; fp32; fp64) -- same bits, timed
    try:
        sing = {"workload": f"replaid.sing on the same {g} x {n} x {m}: colranks(ties = min) + crossprod(rank / nrow - 0.5), "
                            "no normalisation"}
        Sref = None
        for name in ("u16", "f32", "f64"):
            ctx.set_option("ranks_f32", name)
            evs = []
            for it in range(3):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), g, "min", False, 1.0, None)
                    e1.record(stream)
                    ctx.dev_spmm_ranks(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0 / g, -0.5, flags.data_ptr())
                    e2.record(stream)
                evs.append((e0, e1, e2))
            torch.cuda.synchronize()
            rk = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in evs[1:]]))
            cp = float(np.mean([e1.elapsed_time(e2) for _, e1, e2 in evs[1:]]))
            sing[name] = {"colranks_ms": round(rk, 3), "crossprod_ms": round(cp, 3),
                          "scores_per_s": round(float(n) * m / ((rk + cp) * 1e-3), 1),
                          "crossprod": _roof({"u16": "spmm_colquad_u16", "f32": "spmm_colpair_mixed", "f64": "spmm_colpair_f64"}[name],
                                             spmm_alg, cp, _traffic("spmm_colquad_u16", f"{g}xNx{m}", n) if name == "u16" else None)}
            if name == "u16":
                Sref = S.clone()
                if a.cpu_sample > 0:
                    from oracle import c_oracle, fullsize
                    cols, crosses = fullsize.probe_columns(n, m, 64)
                    idx = torch.as_tensor(cols, device=dev)
                    Xc = np_f(X.index_select(0, idx))
                    r_o = c_oracle.colranks_dense_mt(Xc, "min", False, _cpu_threads())
                    assert np.array_equal(np_f(R.index_select(0, idx)), r_o), "min-ties ranks differ from the oracle"
                    s_o = c_oracle.crossprod_dense(r_o / g - 0.5, Gp, Gi, "mean", _cpu_threads())
                    s_g = np_f(S.index_select(0, idx))
                    np.testing.assert_allclose(s_g, s_o, rtol=1e-5, atol=1e-9)
                    sing["parity"] = {"launch": "full", "columns": int(len(cols)), "offsets_past_2^31_checked": bool(crosses),
                                      "ranks_bit_exact": True, "max_abs_err_vs_oracle": float(np.max(np.abs(s_g - s_o)))}
            else:
                sing[name]["bit_identical_to_u16"] = bool(torch.equal(S, Sref))
        sing["speedup_u16_vs_f32_staging"] = round(sing["f32"]["crossprod_ms"] / sing["u16"]["crossprod_ms"], 2)
        sing["speedup_u16_vs_f64"] = round(sing["f64"]["crossprod_ms"] / sing["u16"]["crossprod_ms"], 2)
        del Sref
        out["sing"] = sing
    except Exception as exc:  # pragma: no cover
        out["sing"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
    finally:
        ctx.set_option("ranks_f32", "u16")
        torch.cuda.empty_cache()
    gs.close()
    del X, R, S
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------- the reference's own benchmark shapes
# experiments/benchmark/benchmark-plaid.R:18-35 scores playdata::GSETxGENE (61,459 real gene sets) on 10,000 cells / samples;
# the published wall times (single R process, unknown CPU "p14") are the only numbers the reference holds at any shape:
REF_PUBLISHED = {
    "pbmc3k": {"genes": 12010, "sets": 61459, "sparse": True, "plaid_s": 110.029, "sing_s": 77.353, "scse_s": 69.052,
               "source": "experiments/benchmark/benchmark-pbmc3k@p14.csv:133 (plaid), :132 (replaid.sing), :131 (replaid.scse)"},
    "brca": {"genes": 17713, "sets": 61510, "sparse": False, "plaid_s": 126.416, "sing_s": 110.004, "scse_s": 90.407,
             "source": "experiments/benchmark/benchmark-brca@p14.csv:133 (plaid), :132 (replaid.sing), :131 (replaid.scse)"},
}


def run_ref_shape(a, env, name):
    """plaid() and replaid.sing() at one of the reference's published shapes, on a gene-set collection with the shape of
    its real one (synth.geneset_csc_real: sizes 3 ... 5,000 + an all-genes set, Zipf gene popularity, duplicated sets).
    Resident timing like every other block, the host entry point an R caller waits for beside it, vs_baseline = scores/s
    over the reference's published scores/s (context: unknown CPU, one R thread)."""
    import numpy as np
    torch, dist, ctx, dev, stream = env["torch"], env["dist"], env["ctx"], env["dev"], env["stream"]
    from plaid_amd import synth
    pub = REF_PUBLISHED[name]
    g, m, n, sparse = pub["genes"], pub["sets"], a.ref_samples, pub["sparse"]
    Gp, Gi = synth.geneset_csc_real(g, m)
    z = int(Gp[-1])
    t0 = time.perf_counter()
    gs = ctx.geneset(g, Gp, Gi)
    t_plan = time.perf_counter() - t0
    info = gs.info()
    with torch.cuda.stream(stream):
        if sparse:
            Xp, Xi, Xx, nnz, max_nnz = synth.device_sparse_cells(torch, dev, g, n, 20250616, density=0.07)
        else:
            gen = torch.Generator(device=dev)
            gen.manual_seed(20250617)
            X = torch.randn((n, g), dtype=torch.float64, device=dev, generator=gen) * 2.0 + 8.0
            nnz = g * n
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        red = torch.zeros(4, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    steps = a.block_steps
    ev = Events(torch, stream, steps, 4)

    def crossprod(fl, fused=False):
        if sparse and fused:
            ctx.dev_spmm_csc_fused(gs, Xp.data_ptr(), Xi.data_ptr(), Xx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0, fl, None, None, nnz=nnz)
        elif sparse:
            ctx.dev_spmm_csc(gs, Xp.data_ptr(), Xi.data_ptr(), Xx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, 0.0, fl, None, nnz=nnz)
        else:
            ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, fl)

    def step(k=None):
        with torch.cuda.stream(stream):
            flags.zero_()
            ev.rec(k, 0)
            crossprod(flags.data_ptr(), fused=True)
            ev.rec(k, 1)
            ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
            ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
            ev.rec(k, 2)
            ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
            ev.rec(k, 3)

    elapsed = _timed(torch, dist, False, dev, steps, 2, step)
    snap = _snapshot(torch, S, med, red[0:2], flags, n, m) if a.cpu_sample > 0 else None
    spmm_ms, med_ms, shift_ms = ev.phase_ms(0), ev.phase_ms(1), ev.phase_ms(2)
    scores = float(n) * m
    value = scores / (elapsed / steps)
    ref_rate = pub["sets"] * 10000.0 / pub["plaid_s"]
    scatter = sparse and nnz * 8 < g * n
    alg = (12.0 * nnz + 4.0 * (n + 1) if sparse else 8.0 * g * n) + 4.0 * z + 4.0 * (m + 1) + 8.0 * m * n
    kname = ("spmm_scatter_csc_f64" if scatter else "spmm_colpair_f64<csc>") if sparse else "spmm_colpair_f64"
    out = {
        "workload": f"plaid() at the reference's published shape '{name}': {'sparse' if sparse else 'dense'} {g} genes x {n} "
                    f"{'cells' if sparse else 'samples'}" + (f" ({100.0 * nnz / (g * n):.1f} % stored)" if sparse else "") +
                    f" x {m} gene sets of realistic shape (z={z}; sizes {int(np.diff(Gp).min())}..{int(np.diff(Gp).max())}, "
                    "Zipf gene popularity, an all-genes set, duplicated sets), fp64, inputs resident in HBM",
        "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3), "scores_per_s": round(value, 1),
        "phases_ms": {"crossprod": round(spmm_ms, 4), "col_medians+sum": round(med_ms, 4), "shift": round(shift_ms, 4)},
        "geneset_plan": {"create_s": round(t_plan, 2), "slot_efficiency_one_column": round(z / info["padded_slots"], 4),
                         "slot_efficiency_pair": round(z / info["padded_slots_pair"], 4), "gene_slices": int(info["gene_slices"]),
                         "index_bytes": int(2 * (info["padded_slots"] + info["padded_slots_pair"]))},
        "kernels": {"crossprod": _roof(kname, alg, spmm_ms),
                    "col_medians": _roof("col_medians_stream_kernel", 8.0 * m * n, med_ms),
                    "shift_columns": _roof("shift_columns_kernel", 16.0 * m * n, shift_ms)},
        "reference": {"seconds": pub["plaid_s"], "scores_per_s": round(ref_rate, 1), "source": pub["source"],
                      "note": "the reference's own published wall time of plaid() at this shape (10,000 columns): one R process on "
                              "an unknown CPU ('p14'), host memory in and out -- context, not a same-box comparison"},
        "vs_baseline": round(value / ref_rate, 1),
        "vs_baseline_note": "resident scores/s (like `value`) over the reference's published scores/s; the PCIe-inclusive ratio an "
                            "R caller would see is host_entry.vs_baseline",
    }
    # replaid.sing at the same shape (the reference's second published row): ranks (min ties; a dgCMatrix ranks its zeros too,
    # R/plaid.R:602-609) + the crossprod of the rank matrix, no normalisation
    try:
        with torch.cuda.stream(stream):
            ld = g + (g & 1)
            R = torch.zeros((n, ld), dtype=torch.float64, device=dev)
            Rx = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev) if sparse else None
        evs = []
        for it in range(3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(stream):
                e0.record(stream)
                if sparse:
                    ctx.dev_colranks_csc_dense_nz(Xp.data_ptr(), Xi.data_ptr(), Xx.data_ptr(), g, n, max_nnz, Rx.data_ptr(),
                                                  R.data_ptr(), ld, "min", False, 1.0)
                else:
                    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), ld, "min", False, 1.0, None)
                e1.record(stream)
                ctx.dev_spmm_ranks(gs, R.data_ptr(), ld, n, S.data_ptr(), m, "mean", 1.0 / g, -0.5, None)
                e2.record(stream)
            evs.append((e0, e1, e2))
        torch.cuda.synchronize()
        rk = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in evs[1:]]))
        cp = float(np.mean([e1.elapsed_time(e2) for _, e1, e2 in evs[1:]]))
        sing_rate = scores / ((rk + cp) * 1e-3)
        out["sing"] = {"colranks_ms": round(rk, 3), "crossprod_ms": round(cp, 3), "scores_per_s": round(sing_rate, 1),
                       "reference_seconds": pub["sing_s"], "vs_baseline": round(sing_rate / (pub["sets"] * 10000.0 / pub["sing_s"]), 1),
                       "crossprod": _roof("spmm_colquad_u16", 8.0 * g * n + 4.0 * z + 4.0 * (m + 1) + 8.0 * m * n, cp)}
        if a.cpu_sample > 0:
            from oracle import c_oracle
            idx = torch.arange(0, min(n, 48), device=dev)
            if sparse:
                ph = Xp[:len(idx) + 1].cpu().numpy()
                Xc = np.zeros((g, len(idx)), order="F")
                ih, xh = Xi[:int(ph[-1])].cpu().numpy(), Xx[:int(ph[-1])].cpu().numpy()
                for j in range(len(idx)):
                    Xc[ih[ph[j]:ph[j + 1]], j] = xh[ph[j]:ph[j + 1]]
            else:
                Xc = np_f(X.index_select(0, idx))
            r_o = c_oracle.colranks_dense_mt(Xc, "min", False, _cpu_threads())
            ranks_ok = bool(np.array_equal(np_f(R.index_select(0, idx))[:g], r_o))
            s_o = c_oracle.crossprod_dense(r_o / g - 0.5, Gp, Gi, "mean", _cpu_threads())
            s_g = np_f(S.index_select(0, idx))
            out["sing"]["parity"] = {"columns": int(len(idx)), "ranks_bit_exact": ranks_ok,
                                     "max_abs_err_vs_oracle": float(np.max(np.abs(s_g - s_o))),
                                     "ok": bool(ranks_ok and np.allclose(s_g, s_o, rtol=1e-5, atol=1e-9))}
        del R, Rx
    except Exception as exc:  # pragma: no cover
        out["sing"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
    torch.cuda.empty_cache()
    if a.cpu_sample > 0:
        from oracle import c_oracle, fullsize
        nt = _cpu_threads()
        # parity of the timed launch: probe columns against the oracle (crossprod, medians, shift)
        if sparse:
            sp_, si_, sx_ = _probe_csc(fullsize, Xp, Xi, Xx, snap["cols"])
            raw_o = c_oracle.crossprod_csc(sp_, si_, sx_, g, Gp, Gi, "mean", nt)
        else:
            raw_o = c_oracle.crossprod_dense(np_f(X.index_select(0, snap["idx"])), Gp, Gi, "mean", nt)
        out["parity"] = _check_plaid_launch(torch, stream, snap, raw_o, lambda: crossprod(None), S, n, m)
        # CPU leg: the oracle on a bounded sample, one core
        nc = min(128, n)
        if sparse:
            ph = Xp[:nc + 1].cpu().numpy()
            ih, xh = Xi[:int(ph[-1])].cpu().numpy(), Xx[:int(ph[-1])].cpu().numpy()
            t1 = time.perf_counter()
            Sc = c_oracle.crossprod_csc(ph, ih, xh, g, Gp, Gi, "mean", 1)
        else:
            Xh = np_f(X[:nc])
            t1 = time.perf_counter()
            Sc = c_oracle.crossprod_dense(Xh, Gp, Gi, "mean", 1)
        t2 = time.perf_counter()
        c_oracle.normalize_medians(Sc)
        t3 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(m * nc / (t3 - t1), 1), "unit": "scores/s", "cores": 1, "kind": "port",
                               "sample": f"first {nc} of {n} columns x {m} sets, plain-C oracle: crossprod {t2 - t1:.2f} s + "
                                         f"normalize_medians {t3 - t2:.2f} s", "cpu_count": os.cpu_count()}
        # what an R caller waits for: host buffers in, S back (PCIe-inclusive; never `scores_per_s`)
        try:
            if sparse:
                hp, hi_, hx = Xp.cpu().numpy(), Xi.cpu().numpy(), Xx.cpu().numpy()
                call = lambda: ctx.plaid_csc(hp, hi_, hx, g, Gp, Gi)
            else:
                Xhost = np_f(X)
                call = lambda: ctx.plaid_dense(Xhost, Gp, Gi)
            del S
            torch.cuda.empty_cache()
            call()
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                Sh = call()
                ts.append(time.perf_counter() - t0)
            out["host_entry"] = {"entry": "plaidhip_plaid_csc" if sparse else "plaidhip_plaid_dense", "ms": round(1e3 * min(ts), 1),
                                 "scores_per_s": round(scores / min(ts), 1), "vs_baseline": round(scores / min(ts) / ref_rate, 1),
                                 "speedup_vs_reference_wall_time": round(pub["plaid_s"] * (n / 10000.0) / min(ts), 1),
                                 "note": "pageable host X in, the 4.9 GB score matrix back over PCIe: everything an R caller waits "
                                         "for, against the reference's published wall time of the same call"}
            del Sh
            # replaid.scse (R/plaid.R:155-190), the third call the reference publishes a wall time for at this shape: host
            # entry only (X in, scores back), probe columns against the oracle under the removeLog2 decision the device took
            import scipy.sparse as sps
            from oracle import plaid_oracle as po
            Xarg = sps.csc_matrix((hx, hi_, hp), shape=(g, n)) if sparse else Xhost
            ctx.scse(Xarg, Gp, Gi)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                Ss = ctx.scse(Xarg, Gp, Gi)
                ts.append(time.perf_counter() - t0)
            removed = bool(ctx.last_scse_removed_log2)
            pc = sorted({0, 1, n // 3, n // 2, n - 2, n - 1})
            Gm = sps.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
            rn = [str(k) for k in range(g)]
            exp = po.replaid_scse(Xarg[:, pc], rn, Gm, rn, remove_log2=removed)
            got = Ss[:, pc]
            out["scse"] = {"entry": "plaidhip_scse", "ms": round(1e3 * min(ts), 1), "scores_per_s": round(scores / min(ts), 1),
                           "reference_seconds": pub["scse_s"] * (n / 10000.0),
                           "vs_baseline": round(pub["scse_s"] * (n / 10000.0) / min(ts), 1), "removed_log2": removed,
                           "parity": {"columns": len(pc), "max_rel_err_vs_oracle": _rel_err(got, exp),
                                      "ok": bool(np.allclose(got, exp, rtol=1e-5, atol=1e-9))},
                           "note": "host entry (pageable X in, 4.9 GB of scores back) against the reference's published wall "
                                   "time of replaid.scse at this shape"}
            del Ss
        except Exception as exc:  # pragma: no cover
            out.setdefault("host_entry", {"error": f"{type(exc).__name__}: {str(exc)[:200]}"})
            if "scse_s" in pub and "scse" not in out:
                out["scse"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
    gs.close()
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing launched this process as a rank: start the N ranks ourselves, BEFORE anything here touches the GPU
        return launch_ranks(a, sys.argv[1:])
    if a.dry_run:
        return dry_run(a)
    import torch
    import torch.distributed as dist

    import plaid_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_DUMP_AFTER"):   # debugging aid: every thread's Python stack after so many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DUMP_AFTER"]), exit=True)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    # BENCH_DIST_BACKEND=gloo is a TEST mode for a 1-GPU box: the ranks share device (local_rank mod device count) and the
    # collectives go through gloo, so that the N > 1 code path (sharding, all-reduces, max-over-ranks timing, rank-0 line)
    # runs on real kernels where RCCL cannot (it refuses two ranks on one GPU).  Its numbers mean nothing.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on 1 GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    seen_world = dist.get_world_size() if use_dist else 1      # what the process group reports, not what was asked for
    if seen_world != a.gpus and os.environ.get("BENCH_FORCE_DIST") != "1":
        raise SystemExit(f"bench.py: the process group has {seen_world} ranks, --gpus asked for {a.gpus}")
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(dev_index, stream.cuda_stream)
    env = {"torch": torch, "dist": dist, "ctx": ctx, "dev": dev, "stream": stream, "world": world, "rank": rank,
           "use_dist": use_dist}

    profile_only = a.profile and a.config in ("c3", "c4", "ref", "c3real") and world == 1
    if a.profile:
        a.preheat_steps = 0
    c2 = None if profile_only else run_c2(a, env)
    blocks = {}
    if world == 1 and not use_dist:
        def ref_blocks():
            return {nm: run_ref_shape(a, env, nm) for nm in REF_PUBLISHED}
        for name, fn in (("c3", lambda: run_sparse_ssgsea(a, env, a.c3_cells, "C3", False)), ("c4", lambda: run_c4(a, env)),
                         ("ref_shape", ref_blocks),
                         ("c3_real", lambda: run_sparse_ssgsea(a, env, a.c3_cells, "C3 on a reference-shaped collection", False, True))):
            if a.config in ("all", name) or (name == "ref_shape" and a.config == "ref") or (name == "c3_real" and a.config == "c3real"):
                try:
                    blocks[name] = fn()
                except Exception as exc:  # a failing secondary block must not take the headline line with it
                    blocks[name] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
                    torch.cuda.empty_cache()
    elif a.config in ("all", "c3"):
        try:
            blocks["c5_shard"] = run_sparse_ssgsea(a, env, a.c5_cells_per_gpu, "C5 shard", True)
        except Exception as exc:
            blocks["c5_shard"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}

    if rank == 0 and profile_only:
        out = {"profile_only": True, "config": a.config, "note": "bench.py --profile: only the named block ran (no headline)"}
        out.update(blocks)
    elif rank == 0:
        out = {
            "metric": METRIC,
            "value": round(c2["value"], 1), "unit": "scores/s", "n_gpus": seen_world, "steps": a.steps,
            "warmup": a.warmup, "preheat_steps": a.preheat_steps, "ms_per_step": round(c2["ms_per_step"], 4),
            "ms_per_step_cold": round(c2["ms_per_step_cold"], 4),
            "value_cold": round(c2["config"]["samples_per_gpu"] * c2["config"]["sets"] * world / (c2["ms_per_step_cold"] * 1e-3), 1),
            "cold_note": "ms_per_step_cold / value_cold: the same K steps timed right after the W warm-up steps, BEFORE the "
                         "pre-heat (same process); ms_per_step / value: after `preheat_steps` more untimed steps (sustained clocks)",
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": c2["config"], "roofline": c2["roofline"], "cpu_baseline": c2["cpu_baseline"],
            "phases_ms": c2["phases_ms"], "kernels": c2["kernels"], "pipeline": c2.get("pipeline"), "parity": c2["parity"], "gather": c2["gather"],
            "mixed_precision": c2["mixed_precision"], "host_entry": c2["host_entry"],
        }
        out.update(blocks)
        rs = blocks.get("ref_shape", {})
        if isinstance(rs, dict) and any(isinstance(v, dict) and "vs_baseline" in v for v in rs.values()):
            # `vs_baseline` above stays null: BASELINE.md holds no published number for THIS metric's configuration (20k genes
            # x 10k x 5k).  The reference's only published timings are at its own shapes; the ratios there:
            out["vs_baseline_at_reference_shapes"] = {k: {"plaid": v.get("vs_baseline"), "plaid_host_entry": v.get("host_entry", {}).get("vs_baseline"),
                                                          "sing": v.get("sing", {}).get("vs_baseline"), "scse_host_entry": v.get("scse", {}).get("vs_baseline"), "source": REF_PUBLISHED[k]["source"]}
                                                      for k, v in rs.items() if isinstance(v, dict) and "vs_baseline" in v}
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        emit(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
