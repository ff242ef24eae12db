#!/bin/bash
# Development (GPU box): A/B of prebuilt libraries on ONE box.  crazyara_amd/lib/variants/{a,b,...}.so are copied over the product
# library in turn (ROUNDS times, interleaved) and timed with `bench.py --timed-only`; prints ms per step of every run.
#   usage: bash scripts/ab_libs.sh "old new" [rounds] [extra bench args]
cd "$(dirname "$0")/.."
L=crazyara_amd/lib/libcrazyara_hip.so
cp $L /tmp/ab_base.so
names=${1:-"old new"}; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  for n in $names; do
    cp crazyara_amd/lib/variants/$n.so $L
    line=$(timeout 200 python bench.py --timed-only "$@" 2>/dev/null | tail -1)
    echo "$n round $r: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], d["unit"], "kernel", d["roofline"].get("kernel_ms"))')"
  done
done
cp /tmp/ab_base.so $L
