#!/bin/bash
# round 4, set p: ablations of tower_p8_kernel (each computes wrong results on purpose): 1 no depthwise arithmetic, 2 no expand MFMAs, 4 no project
# MFMAs, 6 neither, 64 no t2 stores, 7 = 1 + 2 + 4
OUT=$(pwd)/gpurun_out/r04p
mkdir -p $OUT
for abl in 1 2 4 6 64 7; do CRA_X3_TOWER=roles scripts/ubench/x3_abl_$abl.bin 256 19 10 1 >> $OUT/ablation_p8.txt 2>&1; done
cat $OUT/ablation_p8.txt
