#!/bin/bash
# Round-2 (second session) evidence with the final binary: GPU tests, smoke, default bench line, kernel
# traces (c4 / c2, forward and forward + backward), counters of the fused step and of the
# linear-Gaussian kernels, PMC traffic of the bench workload's own operands.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
STAGE=${1:-all}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "tests" ]; then
  timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "bench" ]; then
  T0=$(date +%s)
  timeout -k 10 1100 python bench.py > $OUT/g_bench_default.json 2> $OUT/g_bench_default.err
  echo "bench default wall seconds: $(( $(date +%s) - T0 ))"
  for W in c4x2 c4x4 c4s c5h; do
    timeout -k 10 600 python bench.py --workload $W --steps 5 --warmup 2 --extras off --no-cpu-baseline --no-backward > $OUT/g_bench_$W.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/g_bench_$W.json')); r=d['roofline']; print('$W', d['value'], d['ms_per_step'], d['mode'], r['avg_launch_us'], r['frac'])"
  done
fi
prof() {  # name, bench args...
  NAME=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g_prof_$NAME -- \
     python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/g_prof_$NAME.log 2>&1)
  STATS=$(ls $OUT/g_prof_$NAME/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 30 > $OUT/g_rocprof_$NAME.csv
  rm -rf $OUT/g_prof_$NAME $OUT/g_prof_$NAME.log
  head -14 $OUT/g_rocprof_$NAME.csv | cut -c1-130
}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "prof" ]; then
  # TunableOp's picks are made in an unprofiled run first, so the traces hold no tuning trials
  T4=/tmp/aesmc_tuned_c4.csv; T2=/tmp/aesmc_tuned_c2.csv
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $T4 > /dev/null 2>&1
  python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $T2 > /dev/null 2>&1
  prof kernel_stats_c4 --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off --tunableop-file $T4
  prof fwd_bwd_c4 --steps 1 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $T4
  prof kernel_stats_c2 --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --no-backward --extras off --tunableop-file $T2
  prof bwd_c2 --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $T2
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "pmc" ]; then
  bash tools/pmc_lg.sh > /dev/null 2>&1
  cp $OUT/pmc_lg_counters.csv $OUT/g_pmc_lg_counters.csv
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/g_pmcstep_$i -- \
       python $GRAFT_REPO_ROOT/tools/pmc_step.py > $OUT/g_pmcstep_$i.log 2>&1)
    CSV=$(ls $OUT/g_pmcstep_$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$CSV" ] && cp $CSV $OUT/g_pmcstep_set$i.csv
    rm -rf $OUT/g_pmcstep_$i
  done
  python tools/pmc_step_summarize.py $OUT/g_pmc_step_counters.csv $(ls $OUT/g_pmcstep_set*.csv) | head -8
  rm -f $OUT/g_pmcstep_set*.csv $OUT/g_pmcstep_*.log
  cp profiles/pmc_traffic.json $OUT/g_pmc_traffic.json
  for WP in "c4 tuned" "c4 stock" "c2 tuned"; do
    set -- $WP
    for C in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g_pmcw_$C -- \
         python $GRAFT_REPO_ROOT/tools/pmc_workload.py $1 $2 6 > $OUT/g_pmcw_$1_$2_$C.log 2>&1)
      CSV=$(ls $OUT/g_pmcw_$C/*/*counter_collection.csv 2>/dev/null | head -1)
      cp $CSV /tmp/pmcw_$C.csv
      rm -rf $OUT/g_pmcw_$C
    done
    python tools/pmc_workload_summarize.py $1 $2 /tmp/pmcw_FETCH_SIZE.csv /tmp/pmcw_WRITE_SIZE.csv $OUT/g_pmc_traffic.json > /dev/null
    tail -1 $OUT/g_pmcw_$1_$2_WRITE_SIZE.log
  done
  python -c "
import json; t=json.load(open('$OUT/g_pmc_traffic.json'))
for wl,e in t.items():
    for k,v in e.items():
        if k!='calibration': print(wl,k,round(v['hbm_bytes_per_launch']/1e6,1),'MB', v.get('algorithmic_bytes_per_launch'))"
fi
