#!/bin/bash
# round 3, GPU call z: float16x3 stem conv reading the NCHW planes itself -- parity (nets, zero copy, search lanes) and per-op times
OUT=$(pwd)/gpurun_out/r03aa
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_nn_parity_gpu.py tests/test_search_gpu.py -m gpu -q -k "float16x3 or two_role or x3 or onnx or zero_copy or poison or gathered or lanes or unfused or other_trunk" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -5 $OUT/pytest_x3.log
timeout 300 python - <<'PY' 2>&1 | tee $OUT/forward_time.txt
import os, tempfile, sys
sys.path.insert(0, 'tests')
import nn_cases
from crazyara_amd.neuralnetapi import HipAPI
for name, B in (("risev2-19", 256), ("risev33-wdlp", 256)):
    cfg, sd, x = nn_cases.make_case(name)
    d = nn_cases.export_case(tempfile.mkdtemp(), cfg.name, cfg, sd, version="1.0")
    net = HipAPI(0, B, d, "float16x3")
    net.time_forward(50)
    ms = net.time_forward(300)
    print(name, B, "forward ms", ms / 300 if ms > 5 else ms, [(n, round(t, 4)) for n, t in net.time_ops(100)])
    net.close()
PY
