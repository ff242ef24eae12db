#!/bin/bash
# round 4, set ap: VALU instructions the scheduling recipe places behind every MFMA of the EXPAND wave's E + D intervals (tower_p8_kernel<3>): 2 / 3 / 4 (shipped) / 6 / 8
OUT=$(pwd)/gpurun_out/r04aq
mkdir -p $OUT
for rep in 1 2 3; do for v in 2 1 5 10 16; do echo "valu per mfma $v" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_sgb$v.bin 256 19 20 1 >> $OUT/harness.txt 2>&1; done; done
grep -A1 "valu per" $OUT/harness.txt | grep -v "^--" | paste - - | sed 's/CRA_X3_ABL=0  B=256 blocks=19 chunk=128://' | cut -c1-70
