#!/bin/bash
# round 4, set s: float16p8 with e5m2 cross terms in both GEMMs of the tower (high bytes of the f16 split): harness time, parity, forward
OUT=$(pwd)/gpurun_out/r04s
mkdir -p $OUT
for rep in 1 2 3; do CRA_X3_TOWER=roles timeout 120 scripts/ubench/x3_p8b.bin 256 19 20 1 >> $OUT/harness.txt 2>&1; done
cat $OUT/harness.txt
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8" > $OUT/pytest_p8.log 2>&1; tail -25 $OUT/pytest_p8.log
timeout 300 python bench.py --timed-only --precision float16p8 --steps 300 --warmup 30 > $OUT/bench_timed.json 2> $OUT/bench_timed.err; cat $OUT/bench_timed.json
