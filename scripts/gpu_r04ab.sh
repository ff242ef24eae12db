#!/bin/bash
# round 4, set ab: RISEv3.3's 5x5 blocks in tower launches of Precision float16p8 (tower_p8_kernel<5>), first-block gates in the launch: parity, per-op times
OUT=$(pwd)/gpurun_out/r04ab
mkdir -p $OUT
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -x -k "float16p8 and (risev33 or risev2-3 or risev2-13)" > $OUT/pytest_p8.log 2>&1; tail -12 $OUT/pytest_p8.log
timeout 300 python scripts/time_ops_net.py risev33 512 float16p8 > $OUT/ops_risev33.txt 2>&1; cat $OUT/ops_risev33.txt
timeout 300 python scripts/time_ops_net.py risev2-19 256 float16p8 > $OUT/ops_risev2.txt 2>&1; tail -8 $OUT/ops_risev2.txt
