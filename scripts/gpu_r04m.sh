#!/bin/bash
# round 4, set m: Precision float16p8 rebuilt -- EXPAND GEMM on the mixed split, residual stream in the PROJECT waves' registers
OUT=$(pwd)/gpurun_out/r04m
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16p8" > $OUT/pytest_p8.log 2>&1; tail -12 $OUT/pytest_p8.log
timeout 300 python scripts/quick_nn_bench.py 19 256 float16p8,float16x3 > $OUT/quick_p8.log 2>&1; tail -4 $OUT/quick_p8.log
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10.bin 256 19 5 0 > $OUT/trace_x3.txt 2>&1
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10.bin 256 19 5 1 > $OUT/trace_p8.txt 2>&1
head -1 $OUT/trace_x3.txt; head -1 $OUT/trace_p8.txt
