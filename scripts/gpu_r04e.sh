#!/bin/bash
# round 4, set e: which co-resident kernel corrupts value_head_kernel's FC1 accumulators (scripts/value_head_aggressor.py)
OUT=$(pwd)/gpurun_out/r04e
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python scripts/value_head_aggressor.py 20000 > $OUT/aggressor.log 2>&1
tail -12 $OUT/aggressor.log
