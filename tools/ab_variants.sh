# A/B of library builds on the GPU box: bash tools/ab_variants.sh <outdir> <variant> [<variant> ...]
# (variant "" = the product library; others: make -C plaid_amd/csrc variant NAME=x DEFS=... -> libplaidhip_x.so).  Per variant: the
# sparse-crossprod tests, then config 3's launch (16,384 cells x 50,000 sets) plain and with the medians selected inside it.
out=$1; shift
mkdir -p $out
for v in "$@"; do
  lib=$PWD/plaid_amd/csrc/libplaidhip${v:+_$v}.so
  export PLAIDHIP_LIB=$lib
  echo "== variant '${v:-product}'" | tee -a $out/ab.log
  if [ -z "$AB_SKIP_TESTS" ]; then
    python3 -m pytest tests/test_gpu_fused_medians.py tests/test_gpu_refshape.py tests/test_gpu_sparse_ranks.py -m gpu -x -q -k "csc or fused or shard or sparse or token or engine" 2>&1 | tail -1 | tee -a $out/ab.log
  fi
  for f in "" "--fused"; do
    python3 tools/bench_spmm.py --kernel c3 --samples ${AB_SAMPLES:-16384} --sets 50000 --iters 5 $f 2>&1 | grep "^c3" | sort -t' ' -k6 -n | head -2 | sed "s/^/$f /" | tee -a $out/ab.log
  done
done
