#!/bin/bash
# Development (GPU box): time the TOWER launch (Precision float16-3k, where the variants of scripts/build_variant.sh apply) with each
# prebuilt variant library.  usage: run_tower_variants.sh [names...]    env NBLK (19), BATCH (256)
cd "$(dirname "$0")/.."
L=crazyara_amd/lib/libcrazyara_hip.so
cp $L /tmp/variants_base.so
names="$@"; [ -z "$names" ] && names=$(ls crazyara_amd/lib/variants | sed 's/\.so$//')
for n in $names; do
  cp crazyara_amd/lib/variants/$n.so $L
  printf "%-16s " $n
  timeout 120 python scripts/quick_nn_bench.py ${NBLK:-19} ${BATCH:-256} float16-3k 2>&1 | grep "per-op" | sed 's/.*per-op ms: //'
done
cp /tmp/variants_base.so $L
