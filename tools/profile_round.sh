#!/bin/bash
# One-shot evidence run for profiles/ (round 2 layout): the bench line, rocprofv3 kernel stats of the same command,
# PMC passes (separate --pmc runs: FETCH_SIZE / WRITE_SIZE / SQ set) for the dominant kernels of C2, C3 and C4, and the
# micro-benchmarks quoted in DESIGN.md.  Run through gpurun:   gpurun -- 'bash tools/profile_round.sh gpurun_out/r02x'
out=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-sample 0 --no-mixed > $out/stats.log 2>&1
SQSET="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY"
run_pmc() {   # tag, then bench_spmm.py arguments
  tag=$1; shift
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "$SQSET" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_${tag}_$i -- python3 tools/bench_spmm.py "$@" > $out/pmc_${tag}_$i.log 2>&1
  done
}
run_pmc c2 --kernel spmm --iters 3
run_pmc c3 --kernel c3 --samples 8192 --sets 50000 --iters 3
run_pmc c4 --kernel c4 --samples 4096 --sets 50000 --iters 3
python3 tools/bench_spmm.py --kernel c3 --samples 4096 --sets 50000 --iters 3 > $out/c3_4096.log 2>&1
python3 tools/bench_spmm.py --kernel c4 --samples 2048 --sets 50000 --iters 3 > $out/c4_2048.log 2>&1
python3 tools/bench_rank.py > $out/rank.log 2>&1
python3 tools/bench_shift.py > $out/shift.log 2>&1
python3 tools/bench_spmm.py --kernel medians --samples 8192 --sets 50000 --iters 10 > $out/medians_50k.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o /tmp/valu_cost 2>/dev/null && UBENCH_ONLY_PAIR=1 /tmp/valu_cost > $out/ubench_pair_loops.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics.hip -o /tmp/lds_atomics 2>/dev/null && /tmp/lds_atomics > $out/ubench_lds_atomics.txt 2>&1
hipcc -O2 tools/ubench/pcie.cpp -o /tmp/pcie -lpthread 2>/dev/null && /tmp/pcie > $out/ubench_pcie.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in ("c2", "c3", "c4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out + f"/pmc_{tag}_summary.txt", "w") as fh:
        for k, d in sorted(agg.items()):
            if not any(x in k for x in ("spmm", "colranks", "medians", "shift")):
                continue
            fh.write(k + "\n")
            for c, v in sorted(d.items()):
                fh.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
# kernel stats of the bench run: one csv
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    os.replace(f, out + "/bench_kernel_stats.csv")
print(open(out + "/pmc_c2_summary.txt").read()[:1500])
PY
tail -c 600 $out/bench.json; tail -1 $out/c3_4096.log; tail -1 $out/c4_2048.log
