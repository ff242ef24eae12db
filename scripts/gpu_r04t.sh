#!/bin/bash
# round 4, set t: exactness of the 8-bit MFMA (scripts/ubench/mfma8_exactness.hip)
OUT=$(pwd)/gpurun_out/r04t
mkdir -p $OUT
for fmt in 1 0; do for spread in 0 2 6 12 20; do timeout 60 scripts/ubench/mfma8_exactness.bin $fmt $spread 100 >> $OUT/mfma8_exactness.txt 2>&1; done; done
cat $OUT/mfma8_exactness.txt
