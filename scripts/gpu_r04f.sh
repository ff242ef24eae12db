#!/bin/bash
# round 4, set f: (1) does a neighbour's transcendental / LDS / store traffic touch a wave's registers (scripts/ubench/neighbour_vgpr.hip);
# (2) the aggressor scan with the softmax as its own launch
OUT=$(pwd)/gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
for k in -1 0 1 2 3 4 5 6 7; do timeout 120 scripts/ubench/neighbour_vgpr.bin $k 2000 200 400 >> $OUT/neighbour_vgpr.log 2>&1; done
cat $OUT/neighbour_vgpr.log | head -60
CRA_X3_NO_FUSED_SOFTMAX=1 timeout 600 python scripts/value_head_aggressor.py 10000 > $OUT/aggressor_unfused_softmax.log 2>&1
tail -9 $OUT/aggressor_unfused_softmax.log
