#!/bin/bash
# round 4, set x: policy head convs of float16p8 on conv3x3_p8_kernel: parity (a few nets), per-op times, forward
OUT=$(pwd)/gpurun_out/r04x
mkdir -p $OUT
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8 and (risev2-3 or risev2-7 or risev2-19 or flat or wdlp)" > $OUT/pytest_p8.log 2>&1; tail -12 $OUT/pytest_p8.log
timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 > $OUT/bench_timed.json 2> $OUT/bench_timed.err; cat $OUT/bench_timed.json
