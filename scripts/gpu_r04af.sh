#!/bin/bash
# round 4, set af: --dry-ranks rehearsal on the round's last tree (three ranks under torch.distributed.run on the one GPU); a driver-shaped bench line
OUT=$(pwd)/gpurun_out/r04af
mkdir -p $OUT
timeout 600 python bench.py --gpus 3 --dry-ranks --steps 20 --warmup 5 --no-config-legs --no-cpu-baseline --no-dropin-leg > $OUT/bench_dry_ranks.json 2> $OUT/bench_dry_ranks.err; tail -c 700 $OUT/bench_dry_ranks.json; tail -3 $OUT/bench_dry_ranks.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-dropin-leg > $OUT/bench_driver_shaped.json 2> $OUT/bench_driver_shaped.err; head -c 600 $OUT/bench_driver_shaped.json
