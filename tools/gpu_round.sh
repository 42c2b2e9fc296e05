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
python - <<PY
import json
try:
    d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("C2", d["value"], d["ms_per_step"], d["phases_ms"], d["roofline"]["frac"])
    for k in ("c3", "c4"):
        b = d.get(k, {})
        print(k, b.get("ms_per_step"), b.get("scores_per_s"), b.get("phases_ms"), b.get("parity"), b.get("error"))
except Exception as e:
    print("no bench line", e)
PY
