#!/bin/bash
# round 4, set a: bisecting the run-to-run divergence of two-lane float16x3 searches (scripts/lane_divergence.py)
OUT=$(pwd)/gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 1500 python scripts/lane_divergence.py --runs 60 --predicts 300 --out $OUT/lane_divergence.jsonl > $OUT/lane_divergence.log 2>&1
tail -c 6000 $OUT/lane_divergence.log
timeout 300 python -m pytest tests/test_search_gpu.py -m gpu -q -x > $OUT/pytest_search.log 2>&1
tail -3 $OUT/pytest_search.log
