#!/bin/bash
# One-shot evidence run for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC passes
# (separate --pmc runs) for the dominant SpMM kernels, and the secondary configs.  Run through gpurun:
#   gpurun -- 'bash tools/profile_round.sh gpurun_out/r01f'
out=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-sample 0 > $out/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_f64_$tag -- python3 tools/bench_spmm.py --iters 3 > $out/pmc_f64_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_mixed_$tag -- python3 tools/bench_spmm.py --iters 3 --precision mixed > $out/pmc_mixed_$tag.log 2>&1
done
python3 tools/bench_spmm.py --kernel c3 --samples 4096 --sets 50000 --iters 3 > $out/c3.log 2>&1
python3 tools/bench_spmm.py --kernel c4 --samples 2048 --sets 50000 --iters 3 > $out/c4.log 2>&1
python3 tools/bench_spmm.py --kernel c4 --samples 2048 --sets 50000 --iters 3 --precision mixed > $out/c4_mixed.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/pmc_summary.txt").read())
PY
tail -c 2500 $out/bench.json; tail -2 $out/c3.log; tail -2 $out/c4.log; tail -2 $out/c4_mixed.log
