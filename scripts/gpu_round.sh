#!/bin/bash
# GPU-box passes, one table of named sets.  usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag> <set> [<set> ...]
# Results land under gpurun_out/<tag>/ (copied into profiles/ by hand).  The one-off launchers of rounds 3 and 4 (scripts/gpu_r03*.sh,
# gpu_r04*.sh) were the same commands with other arguments; profiles/NOTES.md names the set letters, git history has the files.
#
#   tests       pytest -m gpu (whole suite)                      smoke     __graft_entry__.smoke()
#   newparity   the full-size parity tests of configs 2 / 3 / 5 + lane determinism
#   bench       python bench.py (every leg) + bench_detail.json   benchquick  the NN legs only (no search / game / drop-in legs)
#   trace       rocprofv3 --kernel-trace --stats of bench.py --timed-only, headline mode (+ float16)
#   pmc         SQ / TCC / GRBM counter passes of the headline forward (scripts/prof_forward.py), separate --pmc runs
#   screen      scripts/coresidency_screen.py: round 4's harness net at batch 64, then the four conformant forwards at batch 256
#   rootcause   scripts/value_head_rootcause.py (packed / scalar FC1) + scripts/ubench/neighbour_mfma.bin, every aggressor kind
#   erratum     neighbour_mfma.bin: packed mul / add / fma victims, MFMA kinds, victim and MFMA roles in one workgroup
#   dropin      scripts/dropin_threads.py (the reference's MCTSAgent on HipAPI nets: Threads x simulations)     onetree  scripts/one_tree_sweep.py
#   round       tests smoke bench trace pmc
TAG=${1:-r05}
shift
SETS="$@"
[ -z "$SETS" ] && SETS="round"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
HEADLINE=${HEADLINE:-float16x3}
python __graft_entry__.py > $OUT/build.log 2>&1 || tail -5 $OUT/build.log

pmc_run() { prec=$1; name=$2; shift; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $REPO/scripts/prof_forward.py 19 256 $prec 3 > $OUT/$name.log 2>&1); }

run_set() {
  case $1 in
    tests)
      timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log ;;
    newparity)
      timeout 900 python -m pytest tests/test_nn_parity_gpu.py tests/test_lane_determinism_gpu.py -m gpu -q -k "${NEWPARITY_K:-full_size or determinism}" > $OUT/pytest_newparity.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_newparity.log; tail -6 $OUT/pytest_newparity.log ;;
    bench)
      timeout 1200 python bench.py --detail-out $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
      wc -c $OUT/bench.json; cat $OUT/bench.json; tail -3 $OUT/bench.err ;;
    benchquick)
      timeout 600 python bench.py --no-search --detail-out $OUT/benchquick_detail.json > $OUT/benchquick.json 2> $OUT/benchquick.err
      wc -c $OUT/benchquick.json; cat $OUT/benchquick.json; tail -3 $OUT/benchquick.err ;;
    trace)
      for prec in $HEADLINE float16; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$prec -- python $REPO/bench.py --timed-only --precision $prec --steps 300 --warmup 30 > $OUT/trace_$prec.log 2>&1)
        find $OUT/trace_$prec -type f ! -name "*stats*.csv" -delete
        f=$(find $OUT/trace_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${prec}_kernel_stats.csv && head -8 $f
      done ;;
    pmc)
      prec=$HEADLINE
      pmc_run $prec ${prec}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
      pmc_run $prec ${prec}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
      pmc_run $prec ${prec}_tcc1 FETCH_SIZE TCC_HIT_sum
      pmc_run $prec ${prec}_tcc2 WRITE_SIZE TCC_MISS_sum
      pmc_run $prec ${prec}_grbm GRBM_GUI_ACTIVE
      for p in sq1 sq2 tcc1 tcc2 grbm; do python scripts/pmc_summary.py $OUT/${prec}_$p > $OUT/pmc_${prec}_$p.txt 2>&1; rm -rf $OUT/${prec}_$p; done
      grep -h -A12 "tower_" $OUT/pmc_${prec}_sq2.txt | head -14 ;;
    lds)
      prec=$HEADLINE
      pmc_run $prec ${prec}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
      python scripts/pmc_summary.py $OUT/${prec}_sq2 > $OUT/pmc_${prec}_sq2.txt 2>&1; rm -rf $OUT/${prec}_sq2
      grep -h -A12 "tower_" $OUT/pmc_${prec}_sq2.txt | head -14 ;;
    icache)
      prec=$HEADLINE
      pmc_run $prec ${prec}_ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
      pmc_run $prec ${prec}_ic2 SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
      for p in ic1 ic2; do python scripts/pmc_summary.py $OUT/${prec}_$p > $OUT/pmc_${prec}_$p.txt 2>&1; rm -rf $OUT/${prec}_$p; grep -h -A8 "tower_p8" $OUT/pmc_${prec}_$p.txt | head -9; tail -3 $OUT/${prec}_$p.log; done ;;
    screen)
      timeout 500 python scripts/coresidency_screen.py --batch 64 --configs x3-v2-3 --launches 2000 --out $OUT/screen_batch64.json > $OUT/screen_batch64.txt 2>&1; tail -12 $OUT/screen_batch64.txt
      timeout 900 python scripts/coresidency_screen.py --launches ${SCREEN_LAUNCHES:-1000} --out $OUT/screen.json > $OUT/screen.txt 2>&1; grep -E "RED|RESULT|Error|error" $OUT/screen.txt | head -40 ;;
    rootcause)
      timeout 300 python scripts/value_head_rootcause.py 10000 64 risev2-3 > $OUT/rootcause_probe_packed.txt 2>&1; grep -E "differ|regions|RESULT" $OUT/rootcause_probe_packed.txt | head -8
      ROOTCAUSE_EXTRA_VARIANT=32 timeout 300 python scripts/value_head_rootcause.py 10000 64 risev2-3 > $OUT/rootcause_probe_scalar.txt 2>&1; grep -E "differ|regions|RESULT" $OUT/rootcause_probe_scalar.txt | head -8
      rm -f $OUT/neighbour_mfma.txt
      for form in 0 1 2; do for kind in -1 0 4; do timeout 120 scripts/ubench/neighbour_mfma.bin $kind 3000 8 600 64 $form >> $OUT/neighbour_mfma.txt 2>&1; done; done
      for kind in 1 2 3; do timeout 120 scripts/ubench/neighbour_mfma.bin $kind 3000 8 600 64 0 >> $OUT/neighbour_mfma.txt 2>&1; done
      timeout 120 scripts/ubench/neighbour_mfma.bin 0 3000 8 600 256 0 >> $OUT/neighbour_mfma.txt 2>&1
      grep -E "^victim form|^  [LAS] " $OUT/neighbour_mfma.txt | cut -c1-200 | awk '/^victim/{n=0} {if (n<4) print; n++}' ;;
    erratum)
      # characterisation of the packed-f32 / MFMA-neighbour fault (profiles/NOTES.md round 5): which packed instruction and operand form, which
      # MFMA, and whether a partner wave of the SAME workgroup is enough
      rm -f $OUT/erratum.txt
      for form in 3 4 5 6 7; do for kind in -1 0; do timeout 120 scripts/ubench/neighbour_mfma.bin $kind 3000 8 600 64 $form >> $OUT/erratum.txt 2>&1; done; done
      for kind in 4 5 6 7; do timeout 120 scripts/ubench/neighbour_mfma.bin $kind 3000 8 600 64 6 >> $OUT/erratum.txt 2>&1; done
      for form in 6 7 5 1; do for blocks in 64 256 1024; do timeout 120 scripts/ubench/neighbour_mfma.bin 100 3000 8 600 $blocks $form >> $OUT/erratum.txt 2>&1; done; done
      grep -E "^victim form|^  [LASP] " $OUT/erratum.txt | cut -c130-360 | awk '/aggressor kind/{n=0} {if (n<2) print; n++}' ;;
    dropin)
      timeout 600 python scripts/dropin_threads.py ${HEADLINE} > $OUT/dropin_threads.txt 2>&1; grep -E "^simulations|Error" $OUT/dropin_threads.txt | cut -c1-330 ;;
    onetree)
      timeout 600 python scripts/one_tree_sweep.py ${HEADLINE} > $OUT/one_tree_sweep.txt 2>&1; grep -E "^\{|Error" $OUT/one_tree_sweep.txt | cut -c1-330 ;;
    round)
      for s in tests smoke bench trace pmc; do run_set $s; done ;;
    *) echo "unknown set $1" ;;
  esac
}
for s in $SETS; do echo "=== set $s ==="; run_set $s; done
ls $OUT | head -60
