#!/bin/bash
# round 3, GPU call s: LDS row pad 16 bytes (rows step 4 banks) against 32 bytes (8 banks) in the float16x3 tower kernels
OUT=$(pwd)/gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
H="hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=0 -Icrazyara_amd/csrc/nn scripts/ubench/x3_tower_ablate.hip"
$H -o /tmp/x3_pad8 2>/dev/null &
$H -DCRA_X3_ROWPAD=16 -o /tmp/x3_pad16 2>/dev/null &
$H -DCRA_X3_ROWPAD=24 -o /tmp/x3_pad24 2>/dev/null &
$H -DCRA_X3_TRACE=10 -o /tmp/x3_pad8_trace 2>/dev/null &
wait
{
for v in pad16 pad8 pad24; do for k in roles symmetric; do echo -n "$v $k "; CRA_X3_TOWER=$k /tmp/x3_$v 256 19 20; done; done
for v in pad16 pad8; do echo -n "$v roles "; CRA_X3_TOWER=roles /tmp/x3_$v 1024 19 10; done
CRA_X3_TOWER=roles /tmp/x3_pad8_trace 256 19 5
} > $OUT/x3_time.txt 2>&1
grep "ms per tower" $OUT/x3_time.txt
sed -n "/workgroup 0/,/wave 1:/p" $OUT/x3_time.txt | head -12
sed -n "/workgroup 0/,/workgroup 131/p" $OUT/x3_time.txt | grep -A8 "wave 4:" | head -9
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or two_role or x3" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -5 $OUT/pytest_x3.log
