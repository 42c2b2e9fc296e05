# bench.py with TWO ranks on ONE GPU (BENCH_DIST_BACKEND=gloo: test mode, the numbers mean nothing): the N > 1 code path --
# bench.py starting its own ranks, sharding, the three all-reduces, max-over-ranks timing, the gathers, the rank-0 line -- on
# real kernels where RCCL cannot run (it refuses two ranks on one device).  The C5 shard is cut to 2,048 cells per rank: gloo
# moves CUDA tensors through the host.  Usage (on the GPU box): bash tools/gloo2_check.sh [outdir]
out=${1:-gpurun_out/gloo2}
mkdir -p $out
export BENCH_DIST_BACKEND=gloo BENCH_DUMP_AFTER=300 BENCH_DETAIL=$out/bench_gloo2_detail.json MASTER_PORT=29533
unset WORLD_SIZE RANK LOCAL_RANK
timeout 350 python3 bench.py --gpus 2 --steps 3 --warmup 1 --preheat-steps 0 --c5-cells-per-gpu 2048 --host-gather > $out/bench_gloo2.json 2> $out/bench_gloo2.err
echo rc=$?
grep -n "File \"/root/repo\|Error\|error" $out/bench_gloo2.err | head -20
python3 - <<PY
import json
t = open("$out/bench_gloo2.json").read().strip().splitlines()
line = json.loads(t[-1])
print("line bytes", len(t[-1]), {k: line[k] for k in ("value", "n_gpus", "ms_per_step", "scaling", "parity_ok")})
d = json.load(open("$out/bench_gloo2_detail.json"))
print("gather", json.dumps(d.get("gather"))[:700])
c = d.get("c5_shard", {})
print({k: c.get(k) for k in ("workload", "ms_per_step", "scores_per_s", "error")})
print("c5 gather", json.dumps(c.get("gather"))[:900])
PY
