#!/bin/bash
# GPU session 2 of round 2: parity of everything changed, backward kernels A/B, default bench line,
# kernel trace of the c4 workload, counter passes over the fused step, PMC traffic on bench operands.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/s2_pytest_gpu.txt
tail -5 $OUT/s2_pytest_gpu.txt
timeout 600 python tools/kbench.py c2 c4 > $OUT/s2_kbench.txt 2>&1
grep -i "backward\|==" $OUT/s2_kbench.txt
timeout 600 python tools/stepbench.py c2 c4s > $OUT/s2_stepbench.txt 2>&1
grep "auto\|==" $OUT/s2_stepbench.txt
TUNED=/tmp/aesmc_tuned.csv
timeout 1500 python bench.py --tunableop-file $TUNED > $OUT/s2_bench_default.json 2> $OUT/s2_bench_default.err
tail -3 $OUT/s2_bench_default.err
python - <<PY
import json
d=json.load(open('$OUT/s2_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras')}, indent=1))
for k,v in d.get('extras',{}).items(): print(k, json.dumps(v)[:2500])
PY
# kernel trace of the headline workload (TunableOp picks cached by the run above: no tuning trials)
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s2_prof_c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off \
   --tunableop-file $TUNED > $OUT/s2_prof_c4.log 2>&1)
STATS=$(ls $OUT/s2_prof_c4/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 14 > $OUT/s2_rocprof_kernel_stats_c4.csv && head -24 $OUT/s2_rocprof_kernel_stats_c4.csv
rm -rf $OUT/s2_prof_c4
# hardware counters of the fused step, one set per run
rocprofv3 -L > $OUT/s2_counters_available.txt 2>&1
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/s2_pmcstep_$i -- \
     python $GRAFT_REPO_ROOT/tools/pmc_step.py > $OUT/s2_pmcstep_$i.log 2>&1)
  CSV=$(ls $OUT/s2_pmcstep_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/s2_pmcstep_set$i.csv
  rm -rf $OUT/s2_pmcstep_$i
  tail -2 $OUT/s2_pmcstep_$i.log
done
python tools/pmc_step_summarize.py $OUT/s2_pmc_step_counters.csv $(ls $OUT/s2_pmcstep_set*.csv)
rm -f $OUT/s2_pmcstep_set*.csv
# PMC traffic on the bench workload's own operands
for WP in "c4 tuned" "c4 stock" "c2 tuned"; do
  set -- $WP
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2_pmcw_$C -- \
       python $GRAFT_REPO_ROOT/tools/pmc_workload.py $1 $2 6 > $OUT/s2_pmcw_$1_$2_$C.log 2>&1)
    CSV=$(ls $OUT/s2_pmcw_$C/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$CSV" ] && cp $CSV /tmp/pmcw_$C.csv
    rm -rf $OUT/s2_pmcw_$C
    tail -1 $OUT/s2_pmcw_$1_$2_$C.log
  done
  python tools/pmc_workload_summarize.py $1 $2 /tmp/pmcw_FETCH_SIZE.csv /tmp/pmcw_WRITE_SIZE.csv $OUT/s2_pmc_traffic.json | tail -30
done
