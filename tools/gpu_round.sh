#!/bin/bash
# one GPU-box pass: parity tests, the bench line, kernel stats of the bench run.  usage: tools/gpu_round.sh <tag> [pytest args]
tag=${1:-r02}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q "$@" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -c 600 $out/bench.err
cp bench_detail.json $out/bench_detail.json 2>/dev/null
python3 - <<PY
import json
try:
    line = open("$out/bench.json").read().strip().splitlines()[-1]
    d = json.loads(line)
    print("line bytes", len(line), "C2", d["value"], d["ms_per_step"], d["phases_ms"], d["roofline"]["frac"], d["roofline"].get("lds_frac"), "parity", d["parity_ok"])
    for k, b in d.get("blocks", {}).items():
        print(k, b)
except Exception as e:
    print("no bench line", e)
PY
