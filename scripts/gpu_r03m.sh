#!/bin/bash
# round 3, GPU call m: does starting the PROJECT waves later (mid-interval barrier / wave priorities) put their MFMAs under the depthwise?
OUT=$(pwd)/gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/run_x3_trace.sh $OUT/x3_trace.txt 10
grep "ms per tower" $OUT/x3_trace.txt
for v in "MIDBAR" "EPRIO=3"; do
  sed -n "/==== traced, -DCRA_X3_$v\$/,/wave 1:/p" $OUT/x3_trace.txt | head -14
  sed -n "/==== traced, -DCRA_X3_$v\$/,/workgroup 131/p" $OUT/x3_trace.txt | grep -A8 "wave 4:" | head -9
done
