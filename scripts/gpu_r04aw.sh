#!/bin/bash
# round 4, set aw: determinism stress + the float16x3 search tests on the templated float16x3 tower
OUT=$(pwd)/gpurun_out/r04aw
mkdir -p $OUT
timeout 170 python -m pytest tests/test_lane_determinism_gpu.py tests/test_search_gpu.py -m gpu -q -k "float16x3" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
