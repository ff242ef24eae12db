#!/bin/bash
# round 4, set am: the stem inside the first tower launch as a function of its own (not inlined): parity (two nets), interleaved A/B of the forward
OUT=$(pwd)/gpurun_out/r04am
mkdir -p $OUT
timeout 600 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8 and (risev2-3 or risev33)" > $OUT/pytest_p8.log 2>&1; tail -3 $OUT/pytest_p8.log
for rep in 1 2 3; do
for mode in sep fused; do
  if [ $mode = sep ]; then export CRA_P8_NO_STEM_FUSION=1; else unset CRA_P8_NO_STEM_FUSION; fi
  timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])" >> $OUT/stem_fused_vs_sep.txt
done
done
cat $OUT/stem_fused_vs_sep.txt
