#!/bin/bash
OUT=$(pwd)/gpurun_out/r03r
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/run_x3_trace.sh $OUT/x3_trace.txt 10
grep "ms per tower" $OUT/x3_trace.txt
for a in 16 8 1 24 2 4 6; do echo "== ABL $a"; sed -n "/==== CRA_X3_ABL=$a\$/,/workgroup 131/p" $OUT/x3_trace.txt | sed -n '4,8p'; sed -n "/==== CRA_X3_ABL=$a\$/,/workgroup 131/p" $OUT/x3_trace.txt | grep -A4 "wave 4:" | head -5; done
