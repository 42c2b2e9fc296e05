#!/bin/bash
# scatter-kernel round trip on the GPU box: the sparse crossprod tests on the product library, then the C3 launch of the
# diag library with phase stamps, then the product library's timing (plain and with the medians selected in the launch)
# usage: tools/ab_c3.sh <tag>
tag=${1:-c3}; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_refshape.py tests/test_gpu_fused_medians.py tests/test_gpu_parity.py -m gpu -x -q -k "csc or fused or shard or scatter or sparse" > $out/pytest_sparse.log 2>&1; tail -2 $out/pytest_sparse.log
PLAIDHIP_LIB=$PWD/plaid_amd/csrc/libplaidhip_diag.so python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 4 --stamps > $out/c3_stamps.txt 2>&1
grep "scatter stamps" $out/c3_stamps.txt; tail -1 $out/c3_stamps.txt
python3 tools/bench_spmm.py --kernel c3 --samples 16384 --sets 50000 --iters 5 > $out/c3_16384.txt 2>&1; tail -2 $out/c3_16384.txt
python3 tools/bench_spmm.py --kernel c3 --samples 16384 --sets 50000 --iters 5 --fused > $out/c3_16384_fused.txt 2>&1; tail -2 $out/c3_16384_fused.txt
