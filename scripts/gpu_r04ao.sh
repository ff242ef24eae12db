#!/bin/bash
# round 4, set ao: the launch-structure test of the headline mode
OUT=$(pwd)/gpurun_out/r04ao
mkdir -p $OUT
timeout 300 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "launch_structure" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
