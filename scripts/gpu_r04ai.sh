#!/bin/bash
# round 4, set ai: chained policy head against two launches, interleaved runs
OUT=$(pwd)/gpurun_out/r04ai
mkdir -p $OUT
for rep in 1 2 3 4; do
for mode in two chain; do
  if [ $mode = two ]; then export CRA_P8_NO_HEAD_CHAIN=1; else unset CRA_P8_NO_HEAD_CHAIN; fi
  timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])" >> $OUT/chain_vs_two.txt
done
done
cat $OUT/chain_vs_two.txt
