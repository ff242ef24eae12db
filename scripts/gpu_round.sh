#!/bin/bash
# One GPU-box pass: gpu tests, headline bench, rocprofv3 kernel-trace stats of the bench, PMC passes of the forward.
# usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag> [tests|notests] [pmc|nopmc]
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
if [ "${2:-tests}" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --timed-only --steps 300 --warmup 30 > $OUT/trace.log 2>&1
# the same for Precision float16 (the one-launch forward kernel: the reference-default mode)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f16 -- python $REPO/bench.py --timed-only --precision float16 --steps 300 --warmup 30 > $OUT/trace_f16.log 2>&1
cd $REPO
find $OUT/trace_f16 -type f ! -name "*stats*.csv" -delete
find $OUT/trace -name "*kernel_stats.csv" | head -3
if [ "${3:-pmc}" = "pmc" ]; then
  cd /tmp
  run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 float16 3 > $OUT/$name.log 2>&1; }
  run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES
  run sq3 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
  run tcc1 FETCH_SIZE TCC_HIT_sum
  run tcc2 WRITE_SIZE TCC_MISS_sum
  run grbm GRBM_GUI_ACTIVE
  # the same forward in Precision fp8 (e4m3 GEMM operands): MFMA / LDS / L2 counters and a kernel trace of its own
  run8() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 fp8 3 > $OUT/$name.log 2>&1; }
  run8 fp8_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run8 fp8_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
  run8 fp8_tcc1 FETCH_SIZE TCC_HIT_sum
  # the headline mode, Precision float16x3 (x3.hip): the per-block kernel's MFMA / LDS / L2 counters
  runx() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 float16x3 3 > $OUT/$name.log 2>&1; }
  runx x3_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  runx x3_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
  runx x3_tcc1 FETCH_SIZE TCC_HIT_sum
  runx x3_tcc2 WRITE_SIZE TCC_MISS_sum
  runx x3_grbm GRBM_GUI_ACTIVE
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp8_trace -- python $REPO/scripts/prof_forward.py 19 256 fp8 200 > $OUT/fp8_trace.log 2>&1
  cd $REPO
  ALL="sq1 sq2 sq3 tcc1 tcc2 grbm fp8_sq1 fp8_sq2 fp8_tcc1 x3_sq1 x3_sq2 x3_tcc1 x3_tcc2 x3_grbm"
  for p in $ALL; do python scripts/pmc_summary.py $OUT/$p > $OUT/pmc_$p.txt 2>&1; done
  # raw counter CSVs are large; keep the summaries only
  for p in $ALL; do rm -rf $OUT/$p; done
  find $OUT/fp8_trace -type f ! -name "*stats*.csv" -delete
fi
# keep only the stats csvs of the trace
find $OUT/trace -type f ! -name "*stats*.csv" -delete
ls -R $OUT | head -40
