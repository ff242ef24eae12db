#!/bin/bash
# round 4, set l: what would the mixed split buy in the EXPAND GEMM?  (timing-only ablation: its instruction mix on the existing registers)
OUT=$(pwd)/gpurun_out/r04l
mkdir -p $OUT
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10.bin 256 19 5 0 > $OUT/trace_x3.txt 2>&1
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10_abl128.bin 256 19 5 0 > $OUT/trace_expand_mix.txt 2>&1
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10_abl128.bin 256 19 5 1 > $OUT/trace_expand_mix_and_p8_project.txt 2>&1
head -1 $OUT/trace_x3.txt; head -1 $OUT/trace_expand_mix.txt; head -1 $OUT/trace_expand_mix_and_p8_project.txt
