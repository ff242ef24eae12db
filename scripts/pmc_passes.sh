#!/bin/bash
# usage: pmc_passes.sh <outdir-under-gpurun_out> -- collects PMC passes for the fused forward (separate runs, no tracing mixed in)
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python /root/repo/scripts/prof_forward.py 19 256 float16 3 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | head -20
