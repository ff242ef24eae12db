#!/bin/bash
# round 3, GPU call p: EXPAND role in two passes with tile 0's depthwise inside pass 1's MFMA stream, against the one-pass form
OUT=$(pwd)/gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
H="hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -Icrazyara_amd/csrc/nn scripts/ubench/x3_tower_ablate.hip"
$H -DCRA_X3_EPASS=1 -o /tmp/x3_ep1 2>/dev/null &
$H -o /tmp/x3_ep2 2>/dev/null &
$H -DCRA_X3_EW=2 -o /tmp/x3_ep2_ew2 2>/dev/null &
$H -DCRA_X3_TRACE=10 -o /tmp/x3_ep2_trace 2>/dev/null &
wait
{
for v in ep1 ep2 ep2_ew2; do for bb in 256 1024; do echo -n "$v roles "; CRA_X3_TOWER=roles /tmp/x3_$v $bb 19 20; done; done
echo -n "symmetric "; CRA_X3_TOWER=symmetric /tmp/x3_ep2 256 19 20
CRA_X3_TOWER=roles /tmp/x3_ep2_trace 256 19 5
} > $OUT/x3_time.txt 2>&1
grep "ms per tower" $OUT/x3_time.txt
sed -n "/workgroup 0/,/wave 1:/p" $OUT/x3_time.txt | head -12
sed -n "/workgroup 0/,/workgroup 131/p" $OUT/x3_time.txt | grep -A8 "wave 4:" | head -9
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or two_role or x3" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -5 $OUT/pytest_x3.log
