#!/bin/bash
# round 3, GPU call l: packed-f32 depthwise with the rank-pair tile rows -- parity of every float16x3 path, timeline, bench line
OUT=$(pwd)/gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or two_role or x3" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -8 $OUT/pytest_x3.log
bash scripts/run_x3_trace.sh $OUT/x3_trace.txt 10
grep -A12 "ABL=0" $OUT/x3_trace.txt | head -14
grep -A60 "ABL=0" $OUT/x3_trace.txt | grep -A8 "wave 4:" | head -9
grep "ms per tower" $OUT/x3_trace.txt
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03l/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'), d['roofline'].get('per_op_ms'))
PY
