#!/bin/bash
# Round 6, the weight-port question (VERDICT r05 next #1a): tower_p8_kernel<3> on the RISEv2-19 tower, 256 boards, with the CRA_X3_ABL timing
# switches (x3.hip) -- what does the launch cost when the 8-bit weight images are fetched at half size (512: the stream of a 3-bytes-per-weight
# layout), not at all (1024), no weights at all (16), and the same on the no-arithmetic kernel (7 = no depthwise, no expand / project MFMAs)?
# usage (repo root, GPU box): bash scripts/run_p8_port_ablation.sh [out file]
OUT=${1:-/dev/stdout}
REPO=$(pwd)
mkdir -p /tmp/p8abl
FLAGS=$(python3 -c 'from crazyara_amd import build; print(*build.device_flags())')
ABLS="0 512 1024 16 7 519 1031 23 1 2 4 6"
pids=()
for abl in $ABLS; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 $FLAGS -DCRA_DEVELOPMENT -DCRA_X3_ABL=$abl -I$REPO/crazyara_amd/csrc/nn \
    $REPO/scripts/ubench/x3_tower_ablate.hip -o /tmp/p8abl/abl_$abl 2> /tmp/p8abl/build_$abl.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
{
  echo "tower_p8_kernel<3>, RISEv2-19 tower, 256 boards, random weights (CRA_X3_ABL bits: 1 no depthwise math, 2 no expand MFMAs, 4 no project MFMAs,"
  echo "16 no weight loads, 512 8-bit weight images at half size (3 B / weight stream), 1024 no 8-bit weight loads (2 B / weight stream)); three rounds, interleaved"
  for round in 1 2 3; do
    for abl in $ABLS; do
      if [ -x /tmp/p8abl/abl_$abl ]; then /tmp/p8abl/abl_$abl 256 19 40 1; else echo "build failed for ABL=$abl"; tail -3 /tmp/p8abl/build_$abl.log; fi
    done
  done
} > $OUT 2>&1
