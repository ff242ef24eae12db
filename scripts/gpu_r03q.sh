#!/bin/bash
# round 3, GPU call q: two-role float16x3 tower in half-intervals (both tiles' depthwise inside the expand MFMA streams)
OUT=$(pwd)/gpurun_out/r03t
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
H="hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -Icrazyara_amd/csrc/nn scripts/ubench/x3_tower_ablate.hip"
$H -o /tmp/x3_ip 2>/dev/null &
$H -DCRA_X3_EW=1 -o /tmp/x3_ip_ew2 2>/dev/null &
$H -DCRA_X3_EW=4 -o /tmp/x3_ip_ew8 2>/dev/null &
$H -DCRA_X3_TRACE=10 -o /tmp/x3_ip_trace 2>/dev/null &
wait
{
for v in ip ip_ew2 ip_ew4; do for bb in 256 1024; do echo -n "$v roles "; CRA_X3_TOWER=roles /tmp/x3_$v $bb 19 20; done; done
echo -n "symmetric "; CRA_X3_TOWER=symmetric /tmp/x3_ip 256 19 20
CRA_X3_TOWER=roles /tmp/x3_ip_trace 256 19 5
} > $OUT/x3_time.txt 2>&1
grep "ms per tower" $OUT/x3_time.txt
sed -n "/workgroup 0/,/wave 1:/p" $OUT/x3_time.txt | head -12
sed -n "/workgroup 0/,/workgroup 131/p" $OUT/x3_time.txt | grep -A8 "wave 4:" | head -9
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or two_role or x3" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -5 $OUT/pytest_x3.log
