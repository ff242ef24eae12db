#!/bin/bash
# round 4, set an: the stem inside the first tower launch at batch 8 (the single-go latency case), RISEv2-7
OUT=$(pwd)/gpurun_out/r04an
mkdir -p $OUT
for rep in 1 2; do
for mode in sep fused; do
  if [ $mode = sep ]; then export CRA_P8_NO_STEM_FUSION=1; else unset CRA_P8_NO_STEM_FUSION; fi
  echo "== $mode" >> $OUT/batch8.txt
  timeout 200 python scripts/time_ops_net.py risev2-7 8 float16p8 2>/dev/null | head -7 >> $OUT/batch8.txt
done
done
cat $OUT/batch8.txt
