#!/bin/bash
# round 4, set aj (the round's closing tree: policy head in one launch, clean stdout): full pass on the tree with Precision float16p8 (e5m2 cross terms in both tower GEMMs and in the policy convs) as the headline -- GPU suite, bench line (all legs), kernel traces of the
# headline and of float16x3 / float16, counter sets of the float16p8 and float16x3 towers
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04aj
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp
for prec in float16p8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$prec -- python $REPO/bench.py --timed-only --precision $prec --steps 300 --warmup 30 > $OUT/trace_$prec.log 2>&1
  find $OUT/trace_$prec -type f ! -name "*stats*.csv" -delete
done
run() { prec=$1; name=$2; shift; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 $prec 3 > $OUT/$name.log 2>&1; }
for prec in float16p8; do
  run $prec ${prec}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run $prec ${prec}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
  run $prec ${prec}_tcc1 FETCH_SIZE TCC_HIT_sum
  run $prec ${prec}_tcc2 WRITE_SIZE TCC_MISS_sum
  run $prec ${prec}_grbm GRBM_GUI_ACTIVE
done
cd $REPO
for prec in float16p8; do for p in sq1 sq2 tcc1 tcc2 grbm; do python scripts/pmc_summary.py $OUT/${prec}_$p > $OUT/pmc_${prec}_$p.txt 2>&1; rm -rf $OUT/${prec}_$p; done; done
ls $OUT | head -40
