#!/bin/bash
# round 4, set y: tower_p8_kernel, EXPAND weight window four slabs deep in a block's first interval (x3_deep1) against two (x3_deep0)
OUT=$(pwd)/gpurun_out/r04y
mkdir -p $OUT
for rep in 1 2 3; do for v in 0 1; do echo "deep window $v" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_deep$v.bin 256 19 20 1 >> $OUT/harness.txt 2>&1; done; done
cat $OUT/harness.txt
