#!/bin/bash
# FETCH_SIZE against known byte counts, per access width (tools/ubench/fetch_calib.hip).  On the GPU box:
#     bash tools/fetch_calib.sh gpurun_out/r05t
out=${1:-gpurun_out/fetch_calib}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
./tools/ubench/fetch_calib 8 > $out/fetch_calib_plain.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -- ./tools/ubench/fetch_calib 8 > $out/pmc_$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
nbytes = None
for line in open(out + "/fetch_calib_plain.log"):
    if "reads" in line:
        nbytes = int(line.split("reads")[1].split()[0])
with open(out + "/fetch_calib_summary.txt", "w") as fh:
    fh.write(f"every kernel reads {nbytes} bytes exactly once (buffer >> L2 + Infinity Cache)\n")
    for k in ("seg256_dword", "stream_dword", "stream_x2", "stream_x4"):
        d = agg.get(k, {})
        fs = d.get("FETCH_SIZE")
        fh.write(f"{k:14s} " + "  ".join(f"{c}={v:.6g}" for c, v in sorted(d.items())) +
                 (f"  -> FETCH_SIZE x 1024 / bytes = {fs * 1024 / nbytes:.3f}" if fs and nbytes else "") + "\n")
print(open(out + "/fetch_calib_summary.txt").read())
PY
