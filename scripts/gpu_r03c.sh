#!/bin/bash
# round 3, GPU call c: where the float16x3 tower's time goes (ablation set, both expand tile shapes), the whole GPU test suite,
# SQ counter sets of the float16x3 forward, the arena with its two pools one after the other / at the same time
OUT=$(pwd)/gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
bash scripts/run_x3_ablation.sh $OUT/x3_ablation.txt
cat $OUT/x3_ablation.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
cp gpurun_out/fp8_block_by_block.txt $OUT/ 2>/dev/null
REPO=$(pwd)
cd /tmp
runx() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 float16x3 3 > $OUT/$name.log 2>&1; }
runx x3_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
runx x3_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
runx x3_grbm GRBM_GUI_ACTIVE
cd $REPO
for p in x3_sq1 x3_sq2 x3_grbm; do python scripts/pmc_summary.py $OUT/$p > $OUT/pmc_$p.txt 2>&1; rm -rf $OUT/$p; done
python scripts/game_legs.py > $OUT/game_legs_concurrent.json 2> $OUT/game_legs.err
CRA_ARENA_SERIAL=1 python scripts/game_legs.py > $OUT/game_legs_serial.json 2>> $OUT/game_legs.err
tail -c 600 $OUT/game_legs_concurrent.json; echo; tail -c 600 $OUT/game_legs_serial.json
