#!/bin/bash
OUT=$(pwd)/gpurun_out/r04o
mkdir -p $OUT
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10.bin 256 19 5 1 > $OUT/trace_p8.txt 2>&1
CRA_X3_TOWER=roles scripts/ubench/x3_trace_blk10_abl256.bin 256 19 5 1 > $OUT/trace_p8_quarter_depthwise_on_project.txt 2>&1
head -1 $OUT/trace_p8.txt; head -1 $OUT/trace_p8_quarter_depthwise_on_project.txt
