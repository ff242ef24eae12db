#!/bin/bash
# round 4, set b: the one-launch value head under a storm of concurrent predicts, stage checksums (scripts/lane_divergence.py dbg_*)
OUT=$(pwd)/gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 1200 python scripts/lane_divergence.py --configs dbg_one,dbg_one_own_lds,dbg_one_no_pk,one_no_dbg_storm --out $OUT/lane_divergence.jsonl > $OUT/lane_divergence.log 2>&1
tail -c 5000 $OUT/lane_divergence.log
