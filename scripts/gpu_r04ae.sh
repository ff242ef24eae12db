#!/bin/bash
# round 4, set ae: depthwise taps as v_pk_fma_f32 on the channel pair (x3_pk1) against scalar FMAs (x3_pk0), tower_p8_kernel and tower_x3_roles_kernel
OUT=$(pwd)/gpurun_out/r04ae
mkdir -p $OUT
for rep in 1 2 3; do for v in 0 1; do for p8 in 1 0; do echo "pk taps $v p8=$p8" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_pk$v.bin 256 19 20 $p8 >> $OUT/harness.txt 2>&1; done; done; done
cat $OUT/harness.txt
