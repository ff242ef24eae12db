#!/bin/bash
# PMC + kernel-trace passes of the forward at batch 512 and 1024 (scripts/gpu_round.sh covers batch 256).  usage: pmc_batches.sh <tag> [precision = float16]
TAG=${1:-r02}
PREC=${2:-float16}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for B in 512 1024; do
  run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${PREC}_b${B}_$name -- python $REPO/scripts/prof_forward.py 19 $B $PREC 3 > $OUT/${PREC}_b${B}_$name.log 2>&1; }
  run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
  run tcc1 FETCH_SIZE TCC_HIT_sum
  run tcc2 WRITE_SIZE TCC_MISS_sum
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${PREC}_b${B}_trace -- python $REPO/scripts/prof_forward.py 19 $B $PREC 200 > $OUT/${PREC}_b${B}_trace.log 2>&1
done
cd $REPO
for B in 512 1024; do
  for p in sq1 sq2 tcc1 tcc2; do python scripts/pmc_summary.py $OUT/${PREC}_b${B}_$p > $OUT/pmc_${PREC}_b${B}_$p.txt 2>&1; rm -rf $OUT/${PREC}_b${B}_$p; done
  find $OUT/${PREC}_b${B}_trace -type f ! -name "*kernel_stats.csv" -delete
done
ls $OUT
