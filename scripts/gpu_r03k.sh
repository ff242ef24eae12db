#!/bin/bash
# round 3, GPU call k: phase timeline of the two-role float16x3 tower
OUT=$(pwd)/gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/run_x3_trace.sh $OUT/x3_trace.txt 10
head -120 $OUT/x3_trace.txt
