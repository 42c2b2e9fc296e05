#!/bin/bash
# One-shot evidence run for profiles/ (round 4 layout, round 5 additions).  Run through gpurun:
#     gpurun --timeout 2400 -- 'bash tools/profile_round.sh gpurun_out/r05x'
#  * the bench line (all blocks);
#  * rocprofv3 --kernel-trace --stats of `python3 bench.py --config c2` (the headline alone: no pre-heat, no mixed-precision
#    leg) and of `--profile --config c3 | c4 | ref` (ONLY that block, no C2 headline in the process), so that every tracked
#    kernel_stats CSV has ONE launch shape per kernel row and its average duration can be read against the bench line's
#    HIP-event time;
#  * PMC passes (separate --pmc runs: FETCH_SIZE / WRITE_SIZE / the SQ set / TCC hits / GRBM) for the dominant kernels of
#    C2 (pair kernel), C3 (scatter kernel, fixed-point and fp64 accumulators), C4 (pair kernel at 50k sets, bucket
#    ranker) and of the rank crossprod (quad kernel), summarised per kernel;
#  * the micro-benchmarks quoted in DESIGN.md (single kernels, the plaid step phase by phase, the general weighted crossprod,
#    the access-pattern ceiling of the wave-per-column kernels).
out=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
BENCH_DETAIL=$PWD/$out/bench_detail.json python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -- python3 bench.py --config c2 --preheat-steps 0 --cpu-sample 0 --no-mixed > $out/stats_c2.log 2>&1
for cfg in c3 c4 ref; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -- python3 bench.py --profile --config $cfg --cpu-sample 0 --no-mixed > $out/stats_$cfg.log 2>&1
done
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
run_pmc c3fused --kernel c3 --samples 8192 --sets 50000 --iters 3 --fused
run_pmc c3f64 --kernel c3 --samples 8192 --sets 50000 --iters 3 --scatter-fixed off --scatter-order column
run_pmc c4 --kernel c4 --samples 4096 --sets 50000 --iters 3
run_pmc c4fused --kernel c4 --samples 4096 --sets 50000 --iters 3 --fused
run_pmc sing --kernel sing --samples 4096 --sets 50000 --iters 2
run_pmc c2step --kernel step --iters 3
python3 tools/bench_spmm.py --kernel c3 --samples 4096 --sets 50000 --iters 3 > $out/c3_4096.log 2>&1
for f in "" "--fused"; do python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 4 $f 2>&1 | grep "^c3" | tail -1; done > $out/c3_8192_fused.log 2>&1
python3 tools/bench_spmm.py --kernel c4 --samples 2048 --sets 50000 --iters 3 > $out/c4_2048.log 2>&1
python3 tools/bench_spmm.py --kernel sing --samples 4096 --sets 50000 --iters 3 > $out/sing_4096_50k.log 2>&1
python3 tools/bench_spmm.py --kernel sing --samples 10000 --sets 5000 --iters 3 > $out/sing_10000_5k.log 2>&1
python3 tools/bench_rank.py > $out/rank.log 2>&1
python3 tools/bench_shift.py > $out/shift.log 2>&1
python3 tools/bench_spmm.py --kernel medians --samples 8192 --sets 50000 --iters 10 > $out/medians_50k.log 2>&1
for m in 1000 3000 5000 6000 8000 20000; do python3 tools/bench_spmm.py --kernel medians --samples 10000 --sets $m --iters 10 2>&1 | grep "^medians" >> $out/medians_sizes.log; done
python3 tools/bench_spmm.py --kernel step --iters 10 2>&1 | grep "^step" > $out/step_c2.log
python3 tools/bench_weighted.py > $out/weighted.log 2>&1
python3 tools/bench_rank_long.py > $out/rank_long.log 2>&1
python3 tools/bench_sparse_sing.py > $out/sparse_sing.log 2>&1
for ab in 0 5 8 9 10 13; do echo "ablate $ab (4096 x 50000 sets)"; PLAIDHIP_LIB=plaid_amd/csrc/libplaidhip_diag.so python3 tools/bench_spmm.py --kernel spmm --samples 4096 --sets 50000 --iters 8 --ablate $ab 2>&1 | grep -E "^spmm" | tail -1; done > $out/pair_partial_ablations_c4.log 2>&1
for f in "" "--fused"; do python3 tools/bench_spmm.py --kernel c4 --samples 8192 --sets 50000 --iters 4 $f 2>&1 | grep "^c4" | tail -1; done > $out/c4_8192_fused.log 2>&1
for ab in 0 2 6 7 5; do PLAIDHIP_LIB=plaid_amd/csrc/libplaidhip_diag.so python3 tools/bench_spmm.py --kernel spmm --iters 10 --ablate $ab 2>&1 | grep -E "^spmm|algorithmic" | tail -2; done > $out/pair_ablations.log 2>&1
for ab in 100 101 102 103 104 105 108 109; do echo "ablate $ab"; PLAIDHIP_LIB=plaid_amd/csrc/libplaidhip_diag.so python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 3 --ablate $ab 2>&1 | grep -E "^c3|stamps" | tail -2; done > $out/scatter_ablations.log 2>&1
python3 tools/bench_shift_cast.py 16384 50000 > $out/shift_cast_ab.log 2>&1
[ -x tools/ubench/scatter_ring ] && ./tools/ubench/scatter_ring 16 2 > $out/ubench_scatter_ring.log 2>&1
[ -x tools/ubench/fetch_calib ] && bash tools/fetch_calib.sh $out/fetch_calib > $out/fetch_calib.log 2>&1   # FETCH_SIZE against known byte counts
[ -x tools/ubench/inexact_flag ] && ./tools/ubench/inexact_flag > $out/ubench_inexact_flag.log 2>&1
[ -x tools/ubench/column_stream ] && ./tools/ubench/column_stream 50000 8192 > $out/ubench_column_stream.log 2>&1
[ -x tools/ubench/stream_rw ] && ./tools/ubench/stream_rw 3.2 > $out/ubench_stream_rw.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in ("c2", "c3", "c3fused", "c3f64", "c4", "c4fused", "sing", "c2step"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # kernel durations of the GRBM pass (pass 5), from its own kernel trace: GRBM_GUI_ACTIVE / 8 / duration is the clock the
    # chip held in that very launch, and SQ_LDS_IDX_ACTIVE / CUs over GRBM_GUI_ACTIVE / 8 the share of it the LDS was busy
    for f in glob.glob(out + f"/pmc_{tag}_5/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]]["DURATION_NS_GRBM_PASS"].append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(out + f"/pmc_{tag}_summary.txt", "w") as fh:
        for k, d in sorted(agg.items()):
            if not any(x in k for x in ("spmm", "colranks", "median", "shift", "colmean")):
                continue
            fh.write(k + "\n")
            for c, v in sorted(d.items()):
                fh.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
# kernel stats of the bench runs: rocprofv3's own per-name csv, and one row per LAUNCH SHAPE (tools/kernel_stats_by_shape.py:
# the calibration launches of the fused-medians calls no longer poison the averages of the main launches)
import subprocess
for cfg in ("c2", "c3", "c4", "ref"):
    subprocess.run([sys.executable, "tools/kernel_stats_by_shape.py", out + f"/stats_{cfg}", out + f"/bench_{cfg}_kernel_stats_by_shape.csv"])
    for f in glob.glob(out + f"/stats_{cfg}/**/*kernel_stats.csv", recursive=True):
        os.replace(f, out + f"/bench_{cfg}_kernel_stats.csv")
print(open(out + "/pmc_c2_summary.txt").read()[:1500])
PY
tail -c 2700 $out/bench.json; tail -1 $out/c3_4096.log; tail -1 $out/c4_2048.log; tail -3 $out/sing_4096_50k.log
