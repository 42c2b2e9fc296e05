out=gpurun_out/q11; mkdir -p $out
python3 -m pytest tests -m gpu -x -q -k "csc or sparse or c3 or scatter or stress" > $out/pytest.txt 2>&1
for i in 1 2; do
PLAIDHIP_LIB=$PWD/plaid_amd/csrc/libplaidhip_base.so python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 3 > $out/c3_base_$i.txt 2>&1
python3 tools/bench_spmm.py --kernel c3 --samples 8192 --sets 50000 --iters 3 > $out/c3_new_$i.txt 2>&1
done
tail -1 $out/c3_*.txt; tail -3 $out/pytest.txt
