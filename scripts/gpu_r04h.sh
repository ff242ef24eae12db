#!/bin/bash
# round 4, set h: the shipped (compute-unit-exclusive) one-launch value head -- stress tests, harness controls, aggressor scan; fp8 MFMA
# issue microbenchmark; full GPU suite; driver-shaped bench line; --dry-ranks rehearsal
OUT=$(pwd)/gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 600 python scripts/lane_divergence.py --configs shipped_one,sharing_one,three_roles --runs 100 --predicts 40000 --out $OUT/lane_divergence.jsonl > $OUT/lane_divergence.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r04h/lane_divergence.jsonl"):
    d = json.loads(ln); print(d["name"], d.get("concurrent_predicts"), d.get("searches"), d.get("error", "")[-300:])
PY
CRA_VALUE_HEAD_LDS_PAD=0 timeout 300 python scripts/value_head_aggressor.py 6000 > $OUT/aggressor_shipped.log 2>&1; grep "aggressor op" $OUT/aggressor_shipped.log
timeout 120 scripts/ubench/mix_fp8.bin > $OUT/mix_fp8.log 2>&1; cat $OUT/mix_fp8.log
timeout 900 python -m pytest tests/test_lane_determinism_gpu.py -m gpu -q > $OUT/pytest_determinism.log 2>&1; tail -3 $OUT/pytest_determinism.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_lane_determinism_gpu.py > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_driver_shape.json; tail -5 $OUT/bench.err
timeout 600 python bench.py --gpus 3 --dry-ranks --steps 20 --warmup 5 --no-config-legs --no-cpu-baseline --no-dropin-leg > $OUT/bench_dry_ranks.json 2> $OUT/bench_dry_ranks.err; tail -c 600 $OUT/bench_dry_ranks.json; tail -5 $OUT/bench_dry_ranks.err
