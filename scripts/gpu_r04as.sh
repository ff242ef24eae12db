#!/bin/bash
# round 4, set as: recipe N = 10 in the library: parity subset, forward
OUT=$(pwd)/gpurun_out/r04as
mkdir -p $OUT
timeout 600 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8 and (risev2-3 or risev2-19 or risev33)" > $OUT/pytest_p8.log 2>&1; tail -3 $OUT/pytest_p8.log
for rep in 1 2 3; do timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])" >> $OUT/forward.txt; done
timeout 300 python bench.py --timed-only --precision float16x3 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('float16x3', d['value'], d['ms_per_step'])" >> $OUT/forward.txt
cat $OUT/forward.txt
