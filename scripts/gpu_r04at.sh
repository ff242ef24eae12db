#!/bin/bash
# round 4, set at: every GPU test that runs Precision float16p8 + the determinism stress, on the recipe change
OUT=$(pwd)/gpurun_out/r04at
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -k "float16p8 or p8 or determinism or launch_structure" > $OUT/pytest_p8.log 2>&1; tail -4 $OUT/pytest_p8.log
