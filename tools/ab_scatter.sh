# A/B of scatter-kernel builds (make -C plaid_amd/csrc variant NAME=... DEFS=...): tests of the sparse crossprod, then the C3 launch
for v in "$@"; do
  export PLAIDHIP_LIB=$PWD/plaid_amd/csrc/libplaidhip$v.so
  echo "== variant '$v'"
  python -m pytest tests/test_gpu_refshape.py tests/test_gpu_fused_medians.py -m gpu -x -q -k "csc or fused or shard" 2>&1 | tail -1
  python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 4 --stamps 2>&1 | grep -v "by wave, WG 5" | tail -3
done
