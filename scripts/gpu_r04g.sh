#!/bin/bash
# round 4, set g: the aggressor scan with the conv kernel's bisecting switches (CRA_X3_CONV_DEV) + the neighbour microbenchmark
OUT=$(pwd)/gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
for dev in 1 2 3; do
  CRA_X3_CONV_DEV=$dev timeout 300 python scripts/value_head_aggressor.py 6000 > $OUT/aggressor_dev$dev.log 2>&1
  echo "== CRA_X3_CONV_DEV=$dev"; grep "aggressor op" $OUT/aggressor_dev$dev.log
done
for k in -1 0 2 3 5 7; do timeout 120 scripts/ubench/neighbour_vgpr.bin $k 1000 200 400 >> $OUT/neighbour_vgpr.log 2>&1; done
cat $OUT/neighbour_vgpr.log | head -40
