"""bench.py's output contract, on the CPU: the ONE stdout line stays under 4 KB whatever the blocks hold (the driver
could not parse round 4's ~23 KB line: BENCH_r04.json parsed = null), and `--gpus N` starts its own ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the HIP library at module level)


def _synthetic_full(pad=1):
    long = "x" * (400 * pad)
    roof = {"kernel": "spmm_colpair_f64", "bound": "hbm", "achieved": 2344.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.293,
            "traffic": 3109516288, "algorithmic_bytes": 2002810176, "kernel_ms": 0.8544, "traffic_source": "profiles/x.txt",
            "lds_roof": {"achieved": 72530.1, "peak": 157286.4, "unit": "GB/s", "frac": 0.4611, "bytes": 1, "note": long},
            "fp64_alu_roof": {"achieved": 16.3, "peak": 78.6, "frac": 0.2, "note": long}}
    block = {"workload": long, "steps": 5, "ms_per_step": 41.4, "scores_per_s": 1.2e11,
             "phases_ms": {"sparse_colranks+max": 1.4, "crossprod": 26.0, "col_medians+sum": 1.0, "shift": 13.0},
             "kernels": {"crossprod": dict(roof, kernel="spmm_scatter_csc_f64"), "shift_columns": dict(roof)},
             "cpu_baseline": {"value": 1.4e7, "cores": 1, "kind": "port", "sample": long},
             "parity": {"launch": "full", "columns": 768, "max_abs_err_vs_oracle": 2e-16, "note": long},
             "mfma_backend": {"ms": 1182.0, "achieved": 253.8, "frac": 0.1, "note": long}, "vs_baseline": 1e4}
    return {"metric": bench.METRIC, "value": 4.5e10, "unit": "scores/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "preheat_steps": 60, "ms_per_step": 1.0997, "ms_per_step_cold": 1.18, "value_cold": 4.2e10, "cold_note": long,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2 dense plaid(): 20000 genes x 10000 samples/GPU x 5000 gene sets", "genes": 20000,
                       "samples_per_gpu": 10000, "sets": 5000, "memberships": 697543, "parallelism": "sample-shard x1"},
            "roofline": roof,
            "cpu_baseline": {"value": 8.4e6, "unit": "scores/s", "cores": 1, "kind": "port", "sample": long,
                             "all_cores": {"value": 1.8e7, "cores": 256, "note": long}, "cpu_count": 256},
            "phases_ms": {"spmm": 0.85, "col_medians+sum": 0.1, "shift": 0.13, "normalize_medians": 0.23},
            "kernels": {"col_medians": dict(roof), "shift_columns": dict(roof)},
            "parity": {"launch": "full", "columns": 512, "max_abs_err_vs_oracle": 2e-14, "note": long},
            "gather": None, "mixed_precision": {"note": long}, "host_entry": {"ms": 40.9, "note": long},
            "c3": dict(block), "c4": dict(block), "c3_real": dict(block), "c5_shard": dict(block),
            "ref_shape": {"pbmc3k": dict(block), "brca": dict(block)},
            "vs_baseline_at_reference_shapes": {"pbmc3k": {"note": long}}}


def test_compact_line_is_small_and_carries_the_contract():
    for pad in (1, 40):
        line = bench.compact_line(_synthetic_full(pad))
        assert "\n" not in line and len(line.encode()) < 4096
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, k
        assert d["roofline"]["frac"] == 0.293 and d["roofline"]["lds_frac"] == 0.4611
        assert d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] == 3109516288
        assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and len(d["cpu_baseline"]["sample"]) <= 200
        assert d["config"]["workload"].startswith("C2 dense plaid()") and "model" not in d["config"]
        assert d["parity_ok"] is True
        assert d["blocks"]["c3"]["frac"] == 0.293 and d["blocks"]["c3"]["parity_ok"] is True


def test_compact_line_drops_block_summaries_before_contract_keys():
    full = _synthetic_full()
    full["ref_shape"] = {f"shape{k}": dict(full["c3"]) for k in range(40)}       # far more blocks than fit
    d = json.loads(bench.compact_line(full))
    assert d["roofline"]["frac"] == 0.293 and d["cpu_baseline"]["value"] == 8.4e6
    assert len(json.dumps(d, separators=(",", ":")).encode()) < 4096


def test_failed_parity_shows_in_the_line():
    full = _synthetic_full()
    full["parity"] = {"launch": "full", "ok": False, "error": "mismatch"}
    full["c4"] = {"error": "HipError: out of memory"}
    d = json.loads(bench.compact_line(full))
    assert d["parity_ok"] is False and "error" in d["blocks"]["c4"]


def test_launcher_argv_is_the_drivers_form():
    cmd = bench.launcher_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_gpus_2_starts_its_own_ranks_dry_run(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: two ranks rendezvous over gloo, the max-over-ranks reduction runs,
    rank 0 prints ONE line with n_gpus = 2 (no GPU work in --dry-run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_DETAIL"] = str(tmp_path / "detail.json")
    env["MASTER_PORT"] = "29617"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["data"].startswith("dry-run")
    assert d["ms_per_step"] == 500.0                       # max over ranks of (1 + rank) s over 4 steps
    assert json.load(open(env["BENCH_DETAIL"]))["n_gpus"] == 2


def test_world_size_mismatch_is_an_error():
    """a rank of a 1-rank launch that was asked for 2 GPUs refuses to print a line claiming either"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


def test_a_failing_rank_fails_the_launcher(tmp_path):
    """`bench.py --gpus 2` forwards the exit status of its child launcher: a rank that dies must not end in a zero exit status
    with a stale or missing line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_DETAIL=str(tmp_path / "detail.json"), MASTER_PORT="29619", BENCH_DRY_FAIL_RANK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0, p.stdout[-500:]
    assert not any(ln.startswith('{"metric"') for ln in p.stdout.splitlines())
