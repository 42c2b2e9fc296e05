"""The RCCL (`nccl`) branch of the multi-GPU path on ONE GPU (`pytest -m gpu`).

Every N > 1 run of earlier rounds went through gloo; what an 8-GPU node would execute first -- `init_process_group("nccl",
device_id=...)`, int32 MAX / fp64 SUM all-reduces on device tensors, the gather -- had never run.  Both tests start FRESH
child processes (subprocess, never exec: a process that has touched the GPU must not be replaced):

  * bench.py with BENCH_FORCE_DIST=1: the driver's own N-GPU entry with a one-rank RCCL group (headline block, then the
    sample-shard block whose three scalars go through all-reduces on device tensors);
  * tests/helpers/rccl_one_rank_child.py: sharded_plaid / sharded_ssgsea / sharded_ssgsea_csc / sharded_plaid_csc and
    gather_scores(to="device" | "host") under a one-rank RCCL group, bit-identical to the calls without a group.

Shard axis: the sample columns (R/plaid.R:107, :634-642)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


def _last_json(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_runs_its_rccl_branch_with_one_rank():
    env = _env(29541)
    env["BENCH_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--preheat-steps", "0", "--config", "c2", "--cpu-sample", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line.get("parity_ok", True) is not False


def test_bench_shard_block_all_reduces_over_rccl():
    """the sample-shard block (config 5's per-GPU flow at a small size): max(rX), the flags and {sum, count} of the medians go
    through dist.all_reduce on DEVICE tensors with the nccl backend"""
    env = _env(29543)
    env["BENCH_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--preheat-steps", "0", "--config", "c3", "--c5-cells-per-gpu", "8192", "--big-sets", "20000",
                        "--block-steps", "2", "--cpu-sample", "0", "--no-gather"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 1


def test_sharded_path_over_a_one_rank_rccl_group_equals_no_group():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank_child.py")], cwd=ROOT,
                       env=_env(29545), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    out = _last_json(p.stdout)
    assert out["ok"] and out["backend"] == "nccl" and out["world"] == 1, out
