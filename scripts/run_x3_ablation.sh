#!/bin/bash
# Development: builds scripts/ubench/x3_tower_ablate.hip once per (CRA_X3_NE, CRA_X3_ABL) pair and runs the set on this box's GPU.
# usage (repo root): bash scripts/run_x3_ablation.sh [out file]
OUT=${1:-/dev/stdout}
REPO=$(pwd)
mkdir -p /tmp/x3abl
ABLS="0 1 2 4 6 8 16 32 64 7 15 31 127"
pids=()
for ne in 1 2; do
  for abl in $ABLS; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_NE=$ne -DCRA_X3_ABL=$abl -I$REPO/crazyara_amd/csrc/nn \
      $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/x3abl/ne${ne}_abl_$abl 2> /tmp/x3abl/build_${ne}_$abl.log &
    pids+=($!)
  done
done
for p in "${pids[@]}"; do wait $p; done
{
  echo "tower_x3_kernel ablation, RISEv2-19 tower, 256 boards (CRA_X3_ABL bits: 1 no depthwise math, 2 no expand MFMAs, 4 no project MFMAs,"
  echo "8 no LDS operand reads, 16 no weight loads, 32 no chunk barriers, 64 no t2 stores; CRA_X3_NE = expand channel tiles per wave)"
  for ne in 1 2; do
    for abl in $ABLS; do
      if [ -x /tmp/x3abl/ne${ne}_abl_$abl ]; then echo -n "NE=$ne "; /tmp/x3abl/ne${ne}_abl_$abl 256 19 20; else echo "build failed for NE=$ne ABL=$abl"; tail -3 /tmp/x3abl/build_${ne}_$abl.log; fi
    done
  done
} > $OUT 2>&1
