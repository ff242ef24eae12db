#!/bin/bash
# round 4, set ah: the policy head's two convs in one launch (conv3x3_p8_chain_kernel): parity, forward, A/B against the two launches
OUT=$(pwd)/gpurun_out/r04ah
mkdir -p $OUT
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8 and (risev2-3 or risev2-7 or risev2-19 or risev33 or lichess)" > $OUT/pytest_p8.log 2>&1; tail -6 $OUT/pytest_p8.log
for mode in chain two; do
  if [ $mode = two ]; then export CRA_P8_NO_HEAD_CHAIN=1; else unset CRA_P8_NO_HEAD_CHAIN; fi
  for rep in 1 2; do timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])" >> $OUT/chain_vs_two.txt; done
done
cat $OUT/chain_vs_two.txt
