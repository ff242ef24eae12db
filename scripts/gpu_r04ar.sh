#!/bin/bash
# round 4, set ar: recipe N = 4 (shipped) / 2 / 10, five interleaved rounds on one box
OUT=$(pwd)/gpurun_out/r04ar
mkdir -p $OUT
for rep in 1 2 3 4 5; do for v in 4 2 10; do echo "valu per mfma $v" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_sgb$v.bin 256 19 20 1 >> $OUT/harness.txt 2>&1; done; done
grep -A1 "valu per" $OUT/harness.txt | grep -v "^--" | paste - - | sed 's/CRA_X3_ABL=0  B=256 blocks=19 chunk=128://' | cut -c1-50
