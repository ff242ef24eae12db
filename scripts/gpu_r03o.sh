#!/bin/bash
# round 3, GPU call o: symmetric float16x3 tower with the depthwise inside the project MFMA stream -- parity and time against the two-role kernel
OUT=$(pwd)/gpurun_out/r03o
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -Icrazyara_amd/csrc/nn scripts/ubench/x3_tower_ablate.hip -o /tmp/x3abl0 2>/dev/null
for k in roles symmetric; do echo -n "$k "; CRA_X3_TOWER=$k /tmp/x3abl0 256 19 20; done | tee $OUT/x3_time.txt
for k in roles symmetric; do echo -n "$k "; CRA_X3_TOWER=$k /tmp/x3abl0 1024 19 10; done | tee -a $OUT/x3_time.txt
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or two_role or x3" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -8 $OUT/pytest_x3.log
for k in roles symmetric; do
CRA_X3_TOWER=$k timeout 300 python - <<'PY'
import os, tempfile, sys
sys.path.insert(0, 'tests')
import nn_cases
from crazyara_amd.neuralnetapi import HipAPI
cfg, sd, x = nn_cases.make_case("risev2-19")
d = nn_cases.export_case(tempfile.mkdtemp(), cfg.name, cfg, sd, version="1.0")
net = HipAPI(0, 256, d, "float16x3")
net.time_forward(50)
ms = net.time_forward(300)
print(os.environ["CRA_X3_TOWER"], "forward ms", ms / 300 if ms > 5 else ms, [(n, round(t, 4)) for n, t in net.time_ops(100)])
PY
done 2>&1 | tee $OUT/forward_time.txt
