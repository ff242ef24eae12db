#!/bin/bash
# round 4, set ag: stdout of bench.py carries the JSON line only (library chatter to stderr): --dry-ranks rehearsal and a driver-shaped line
OUT=$(pwd)/gpurun_out/r04ag
mkdir -p $OUT
timeout 600 python bench.py --gpus 3 --dry-ranks --steps 20 --warmup 5 --no-config-legs --no-cpu-baseline --no-dropin-leg > $OUT/bench_dry_ranks.json 2> $OUT/bench_dry_ranks.err; wc -l $OUT/bench_dry_ranks.json; head -c 300 $OUT/bench_dry_ranks.json; echo; tail -3 $OUT/bench_dry_ranks.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-dropin-leg --no-cpu-baseline > $OUT/bench_torchrun_1.json 2> $OUT/bench_torchrun_1.err; wc -l $OUT/bench_torchrun_1.json; head -c 200 $OUT/bench_torchrun_1.json
