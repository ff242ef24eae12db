#!/bin/bash
# round 4, set ac: the lane step of a float16p8 search (5 launches per forward): replayed graph (shipped) against plain stream launches
OUT=$(pwd)/gpurun_out/r04ac
mkdir -p $OUT
for rep in 1 2; do
for mode in graph plain; do
  if [ $mode = plain ]; then export CRA_LANE_NO_GRAPH=1; else unset CRA_LANE_NO_GRAPH; fi
  timeout 300 python bench.py --no-cpu-baseline --no-config-legs --no-dropin-leg --no-live-pmc --search-precision-other float16p8 2> $OUT/err_$mode.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['mcts']
print('$mode', 'nn', d['value'], 'config2 nodes/s', m['mcts_nodes_per_sec'], m['nodes_per_sec_repeats'], 'by threads', m.get('nodes_per_sec_by_host_threads'))" >> $OUT/lane_graph_vs_plain.txt
done
done
cat $OUT/lane_graph_vs_plain.txt
