#!/bin/bash
# Development: builds scripts/ubench/x3_tower_ablate.hip once per CRA_X3_ABL switch and runs the set on this box's GPU, for the
# symmetric kernel (CRA_X3_TOWER=symmetric) and the two-role kernel (default).
# usage (repo root): bash scripts/run_x3_ablation.sh [out file]
# every device compile takes the library's flags (no packed f32 arithmetic: crazyara_amd/build.py)
FLAGS=$(cd "$(dirname "$0")/.." && python3 -c 'from crazyara_amd import build; print(*build.device_flags())')
OUT=${1:-/dev/stdout}
REPO=$(pwd)
mkdir -p /tmp/x3abl
ABLS="0 1 2 4 6 8 16 32 7 15 31 127"
pids=()
for abl in $ABLS; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 $FLAGS -DCRA_DEVELOPMENT -DCRA_X3_ABL=$abl -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/x3abl/abl_$abl 2> /tmp/x3abl/build_$abl.log &
  pids+=($!)
done
for v in EW=4 PW=4 "EW=4 -DCRA_X3_PW=4"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 $FLAGS -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -DCRA_X3_$v -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o "/tmp/x3abl/var_${v// /_}" 2> "/tmp/x3abl/build_var_${v// /_}.log" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
{
  echo "tower_x3 kernels, RISEv2-19 tower, 256 boards (CRA_X3_ABL bits: 1 no depthwise math, 2 no expand MFMAs, 4 no project MFMAs,"
  echo "8 no LDS operand reads, 16 no weight loads, 32 no chunk barriers, 64 no t2 stores)"
  for kern in roles symmetric; do
    for abl in $ABLS; do
      if [ -x /tmp/x3abl/abl_$abl ]; then echo -n "$kern "; CRA_X3_TOWER=$kern /tmp/x3abl/abl_$abl 256 19 20; else echo "build failed for ABL=$abl"; tail -3 /tmp/x3abl/build_$abl.log; fi
    done
  done
  for v in EW=4 PW=4 "EW=4 -DCRA_X3_PW=4"; do echo -n "roles, -DCRA_X3_$v: "; CRA_X3_TOWER=roles "/tmp/x3abl/var_${v// /_}" 256 19 20; done
  for bb in 512 1024; do echo -n "roles "; CRA_X3_TOWER=roles /tmp/x3abl/abl_0 $bb 19 10; echo -n "symmetric "; CRA_X3_TOWER=symmetric /tmp/x3abl/abl_0 $bb 19 10; done
} > $OUT 2>&1
