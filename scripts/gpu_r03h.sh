#!/bin/bash
# round 3, GPU call h: L2 warm-up touches for the weight stream of the two-role tower
OUT=$(pwd)/gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_nn_parity_gpu.py -m gpu -q -k "float16x3 or onnx or headline" > $OUT/pytest_x3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_x3.log
tail -8 $OUT/pytest_x3.log
bash scripts/run_x3_ablation.sh $OUT/x3_ablation.txt
cat $OUT/x3_ablation.txt
timeout 300 python bench.py --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench_timed_only.json
CRA_X3_TOWER=symmetric timeout 300 python bench.py --timed-only > $OUT/bench_timed_only_symmetric.json 2>> $OUT/bench.err
tail -c 700 $OUT/bench_timed_only_symmetric.json
REPO=$(pwd)
cd /tmp
runx() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 float16x3 3 > $OUT/$name.log 2>&1; }
runx x3_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
runx x3_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
runx x3_grbm GRBM_GUI_ACTIVE
cd $REPO
for p in x3_sq1 x3_sq2 x3_grbm; do python scripts/pmc_summary.py $OUT/$p tower_x3 > $OUT/pmc_$p.txt 2>&1; rm -rf $OUT/$p; done
cat $OUT/pmc_x3_sq1.txt $OUT/pmc_x3_sq2.txt $OUT/pmc_x3_grbm.txt
