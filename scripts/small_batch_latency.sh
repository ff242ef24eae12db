#!/bin/bash
# Round 6: forward latency of the small-batch (split-board) path next to the one-workgroup-per-board tower, RISEv2-19 at batch 1 / 8 / 32 and
# RISEv2-7 at batch 8 (BASELINE config 1).  usage (repo root, GPU box): bash scripts/small_batch_latency.sh <out file> [precisions]
OUT=${1:-/dev/stdout}
PRECS=${2:-float16p8,float16p8-1wg}
{
  for B in 1 8 16 32; do python scripts/quick_nn_bench.py 19 $B $PRECS; done
  python scripts/quick_nn_bench.py 7 8 $PRECS
} 2>&1 | grep -v amdgpu.ids > $OUT
