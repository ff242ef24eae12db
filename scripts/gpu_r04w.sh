#!/bin/bash
# round 4, set w: depthwise edge masks as address offsets (x3_p8c) against the previous build (x3_p8b_prev); both towers; parity of the x3 / p8 modes
OUT=$(pwd)/gpurun_out/r04w
mkdir -p $OUT
for rep in 1 2 3; do
  for v in b_prev c; do
    for p8 in 1 0; do echo "kernel $v p8=$p8" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_p8$v.bin 256 19 20 $p8 >> $OUT/harness.txt 2>&1; done
  done
done
cat $OUT/harness.txt
timeout 1500 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16p8 or float16x3" > $OUT/pytest_x3.log 2>&1; tail -8 $OUT/pytest_x3.log
