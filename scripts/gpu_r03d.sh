#!/bin/bash
# round 3, GPU call d: the float16x3 tower with buffer-load weight windows + fenced double-buffered LDS fragments: parity, ablation, rate
OUT=$(pwd)/gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or onnx or headline" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -8 $OUT/pytest_x3.log
bash scripts/run_x3_ablation.sh $OUT/x3_ablation.txt
cat $OUT/x3_ablation.txt
timeout 300 python bench.py --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench_timed_only.json
