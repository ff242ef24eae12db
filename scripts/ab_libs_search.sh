#!/bin/bash
# Development (GPU box): like ab_libs.sh, for the main search leg (BASELINE config 2) of bench.py.
#   usage: bash scripts/ab_libs_search.sh "a b" [rounds]
cd "$(dirname "$0")/.."
L=crazyara_amd/lib/libcrazyara_hip.so
cp $L /tmp/ab_base.so
names=${1:-"old new"}; rounds=${2:-2}
for r in $(seq $rounds); do
  for n in $names; do
    cp crazyara_amd/lib/variants/$n.so $L
    line=$(timeout 200 python bench.py --no-cpu-baseline --no-config-legs --no-live-pmc --steps 100 --warmup 20 2>/dev/null | tail -1)
    echo "$n round $r: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); m=d["mcts"]; print("nn", d["value"], "mcts", m["mcts_nodes_per_sec"], m["nodes_per_sec_repeats"], "fill", m["avg_batch_fill"], "rounds", m["rounds"])')"
  done
done
cp /tmp/ab_base.so $L
