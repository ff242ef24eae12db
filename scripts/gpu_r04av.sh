#!/bin/bash
# round 4, set au: RISEv3.3's 5x5 blocks in tower launches of Precision float16x3 too (tower_x3_roles_kernel<5>): parity, per-op times
OUT=$(pwd)/gpurun_out/r04av
mkdir -p $OUT
timeout 600 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "(risev33 or risev2-3) and float16x3" > $OUT/pytest_x3.log 2>&1; tail -4 $OUT/pytest_x3.log

