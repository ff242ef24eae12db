#!/bin/bash
# round 4, set aa: where RISEv3.3's forward goes in the conformant modes (per-op event times)
OUT=$(pwd)/gpurun_out/r04aa
mkdir -p $OUT
for prec in float16p8 float16; do timeout 300 python scripts/time_ops_net.py risev33 512 $prec >> $OUT/ops_risev33.txt 2>&1; done
cat $OUT/ops_risev33.txt
