#!/bin/bash
# round 3, GPU call j: window depth / MFMA order variants of the two-role float16x3 tower, its measured errors, the new GPU tests
# (two-role == symmetric bit for bit, float16x3 search lanes), the game legs with the lighter node arena
OUT=$(pwd)/gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_nn_parity_gpu.py tests/test_search_gpu.py tests/test_selfplay.py -m gpu -q -k "two_role or gathered or selfplay or headline" > $OUT/pytest_sel.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_sel.log
tail -6 $OUT/pytest_sel.log
bash scripts/run_x3_ablation.sh $OUT/x3_ablation.txt
grep -v "^symmetric CRA_X3_ABL=[1-9]\|^roles CRA_X3_ABL=[1-9]" $OUT/x3_ablation.txt
timeout 600 python scripts/f16_error_scan.py float16x3 float16x3-perblock > $OUT/x3_error_scan.txt 2>&1
cat $OUT/x3_error_scan.txt
python scripts/game_legs.py > $OUT/game_legs.json 2> $OUT/game_legs.err
tail -c 900 $OUT/game_legs.json
