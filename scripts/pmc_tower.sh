#!/bin/bash
# PMC passes of the forward, summarised for one kernel-name filter.  usage: pmc_tower.sh <tag> [filter]
TAG=${1:-t}
FILT=${2:-tower}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 float16 3 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES
run sq3 SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
cd $REPO
for p in sq1 sq2 sq3 tcc1 tcc2; do python scripts/pmc_summary.py $OUT/$p $FILT > $OUT/pmc_$p.txt 2>&1; rm -rf $OUT/$p; done
cat $OUT/pmc_*.txt
