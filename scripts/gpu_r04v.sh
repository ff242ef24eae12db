#!/bin/bash
# round 4, set v: float16p8 (e5m2, both GEMMs): every GPU test that runs the mode + the determinism stress + a forward
OUT=$(pwd)/gpurun_out/r04v
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -k "float16p8 or p8 or determinism" > $OUT/pytest_p8.log 2>&1; tail -25 $OUT/pytest_p8.log
timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 > $OUT/bench_timed.json 2> $OUT/bench_timed.err; cat $OUT/bench_timed.json
