#!/bin/bash
# Development: phase timeline of the two-role float16x3 tower (CRA_X3_TRACE stamps, x3.hip) for one block, full kernel and with single
# parts switched off (CRA_X3_ABL).   usage (repo root): bash scripts/run_x3_trace.sh [out file] [block]
OUT=${1:-/dev/stdout}
BLK=${2:-10}
REPO=$(pwd)
mkdir -p /tmp/x3trace
pids=()
for abl in 0 16 8 1 24; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=$abl -DCRA_X3_TRACE=$BLK -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/x3trace/t_$abl 2> /tmp/x3trace/build_$abl.log &
  pids+=($!)
done
for v in MIDBAR EPRIO=3 "EPRIO=3 -DCRA_X3_EW=4"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -DCRA_X3_TRACE=$BLK -DCRA_X3_$v -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o "/tmp/x3trace/v_${v// /_}" 2> "/tmp/x3trace/build_v_${v// /_}.log" &
  pids+=($!)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -DCRA_X3_$v -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o "/tmp/x3trace/n_${v// /_}" 2> "/tmp/x3trace/build_n_${v// /_}.log" &
  pids+=($!)
done
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -I$REPO/crazyara_amd/csrc/nn $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/x3trace/n_base 2> /tmp/x3trace/build_n_base.log &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
{
  echo "==== untraced builds"
  echo -n "base: "; CRA_X3_TOWER=roles /tmp/x3trace/n_base 256 19 20
  for v in MIDBAR EPRIO=3 "EPRIO=3 -DCRA_X3_EW=4"; do echo -n "-DCRA_X3_$v: "; CRA_X3_TOWER=roles "/tmp/x3trace/n_${v// /_}" 256 19 20; done
  for v in MIDBAR EPRIO=3 "EPRIO=3 -DCRA_X3_EW=4"; do echo "==== traced, -DCRA_X3_$v"; CRA_X3_TOWER=roles "/tmp/x3trace/v_${v// /_}" 256 19 5; done
  for abl in 0 16 8 1 24; do
    echo "==== CRA_X3_ABL=$abl"
    CRA_X3_TOWER=roles /tmp/x3trace/t_$abl 256 19 5
  done
} > $OUT 2>&1
