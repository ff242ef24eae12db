#!/bin/bash
# round 4, set r: tower_p8_kernel with the weight stream requested ahead at the block boundaries / in the steady state (CRA_X3_PF bits, x3.hip)
OUT=$(pwd)/gpurun_out/r04r
mkdir -p $OUT
for rep in 1 2; do
for pf in 0 1 3 8 11; do echo "CRA_X3_PF=$pf" >> $OUT/pf_variants.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/pf/pf_$pf 256 19 20 1 >> $OUT/pf_variants.txt 2>&1; done
done
cat $OUT/pf_variants.txt
