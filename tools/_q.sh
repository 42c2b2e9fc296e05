out=gpurun_out/q19; mkdir -p $out
python3 -m pytest tests -m gpu -x -q -k "rank or ssgsea or csc" > $out/pytest.txt 2>&1
python3 bench.py --config c3 --cpu-sample 0 > $out/c3.json 2> $out/c3.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/q19/c3.json").read().strip().splitlines()[-1])
print((d.get("c3") or {}).get("phases_ms"))
PY
tail -n 3 $out/pytest.txt
