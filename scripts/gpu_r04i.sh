#!/bin/bash
# round 4, set i: Precision float16p8 (project cross terms on e4m3 MFMAs), first hardware contact: the scale operand's sense, parity, time
OUT=$(pwd)/gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
for ls in 0.00048828125 2048; do
  echo "== CRA_P8_LO_SCALE=$ls"
  CRA_P8_LO_SCALE=$ls timeout 600 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8_equals" > $OUT/pytest_p8_scale_$ls.log 2>&1; tail -15 $OUT/pytest_p8_scale_$ls.log
done
timeout 300 python scripts/quick_nn_bench.py 19 256 float16p8,float16x3 > $OUT/quick_p8.log 2>&1; tail -12 $OUT/quick_p8.log
