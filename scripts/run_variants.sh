#!/bin/bash
# Development (GPU box): time the forward with each prebuilt tower variant (scripts/build_variant.sh).  Usage: run_variants.sh [names...]
cd "$(dirname "$0")/.."
L=crazyara_amd/lib/libcrazyara_hip.so
cp $L crazyara_amd/lib/variants/base.so
names="$@"; [ -z "$names" ] && names=$(ls crazyara_amd/lib/variants | sed 's/\.so$//')
for n in $names; do
  cp crazyara_amd/lib/variants/$n.so $L
  echo "=== $n"
  timeout 120 python scripts/quick_nn_bench.py ${NBLK:-19} ${BATCH:-256} float16 2>&1 | grep -v amdgpu.ids | tail -2
  if [ -n "$TRACE" ]; then CRA_TOWER_TRACE=1 timeout 120 python scripts/quick_nn_bench.py ${NBLK:-19} ${BATCH:-256} float16 2>&1 | grep "tower trace"; fi
done
cp crazyara_amd/lib/variants/base.so $L
