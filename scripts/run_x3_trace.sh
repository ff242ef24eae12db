#!/bin/bash
# Development: phase timeline of the two-role float16x3 tower (CRA_X3_TRACE stamps, x3.hip) for one block, full kernel and with single
# parts switched off (CRA_X3_ABL).   usage (repo root): bash scripts/run_x3_trace.sh [out file] [block]
# every device compile takes the library's flags (no packed f32 arithmetic: crazyara_amd/build.py)
FLAGS=$(cd "$(dirname "$0")/.." && python3 -c 'from crazyara_amd import build; print(*build.device_flags())')
OUT=${1:-/dev/stdout}
BLK=${2:-10}
REPO=$(pwd)
mkdir -p /tmp/x3trace
pids=()
for abl in 0 16 8 1 24 2 4 6; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 $FLAGS -DCRA_DEVELOPMENT -DCRA_X3_ABL=$abl -DCRA_X3_TRACE=$BLK -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/x3trace/t_$abl 2> /tmp/x3trace/build_$abl.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
{
  for abl in 0 16 8 1 24 2 4 6; do
    echo "==== CRA_X3_ABL=$abl"
    CRA_X3_TOWER=roles /tmp/x3trace/t_$abl 256 19 5
  done
} > $OUT 2>&1
