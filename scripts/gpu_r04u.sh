#!/bin/bash
# round 4, set u: tower_p8_kernel old (e4m3, expand GEMM only: x3_p8a) against new (e5m2 high bytes, both GEMMs: x3_p8b) on one box; timeline
# of block 10 and the ablations of the new kernel
OUT=$(pwd)/gpurun_out/r04u
mkdir -p $OUT
for rep in 1 2 3; do
  for v in a b; do echo "kernel $v" >> $OUT/harness.txt; CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_p8$v.bin 256 19 20 1 >> $OUT/harness.txt 2>&1; done
done
cat $OUT/harness.txt
CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_p8b_trace.bin 256 19 5 1 > $OUT/trace_p8b.txt 2>&1
for abl in 1 2 4 6 7; do CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_p8b_abl_$abl.bin 256 19 10 1 >> $OUT/ablation_p8b.txt 2>&1; done
cat $OUT/ablation_p8b.txt
