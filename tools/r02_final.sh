#!/bin/bash
# Regenerates the round's evidence with the final binary: GPU tests, default bench line, traces (c4 and
# c2, forward; c4 and c2 with backward), counters of the fused step, shard costs of the strong curve.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
TUNED=/tmp/aesmc_tuned.csv
T0=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 --tunableop-file $TUNED > $OUT/f_bench_default.json 2> $OUT/f_bench_default.err
echo "bench default (--steps 20 --warmup 5) wall seconds: $(( $(date +%s) - T0 ))"
prof() {  # name, bench args...
  NAME=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f_prof_$NAME -- \
     python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/f_prof_$NAME.log 2>&1)
  STATS=$(ls $OUT/f_prof_$NAME/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 30 > $OUT/f_rocprof_$NAME.csv
  rm -rf $OUT/f_prof_$NAME
  grep "aesmc::ancestor_index_inv\|aesmc::normal_logweight_kernel<float; 23\|bwd" $OUT/f_rocprof_$NAME.csv | cut -c1-120
}
prof kernel_stats_c4 --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off --tunableop-file $TUNED
prof fwd_bwd_c4 --steps 1 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $TUNED
TUNED2=/tmp/aesmc_tuned_c2.csv
python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $TUNED2 > /dev/null 2>&1
prof kernel_stats_c2 --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --no-backward --extras off --tunableop-file $TUNED2
prof bwd_c2_tuned --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $TUNED2
for W in c4x2 c4x4 c4s; do
  timeout 900 python bench.py --workload $W --steps 5 --warmup 2 --extras off --no-cpu-baseline --no-backward --tunableop-file /tmp/tuned_$W.csv > $OUT/f_bench_$W.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/f_bench_$W.json')); r=d['roofline']; print('$W', d['value'], d['ms_per_step'], d['mode'], r['avg_launch_us'], r['frac'])"
done
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/f_pmcstep_$i -- \
     python $GRAFT_REPO_ROOT/tools/pmc_step.py > $OUT/f_pmcstep_$i.log 2>&1)
  CSV=$(ls $OUT/f_pmcstep_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/f_pmcstep_set$i.csv
  rm -rf $OUT/f_pmcstep_$i
done
python tools/pmc_step_summarize.py $OUT/f_pmc_step_counters.csv $(ls $OUT/f_pmcstep_set*.csv) | head -8
rm -f $OUT/f_pmcstep_set*.csv $OUT/f_pmcstep_*.log
python - <<PY
import json
d=json.load(open('$OUT/f_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras','config','cpu_baseline')}, indent=None)[:1500])
e=d['extras']; c2=e['c2_hipgraph']
print('c2', c2['value'], c2['ms_per_step'], c2['fwd_bwd_particle_steps_per_sec'], c2['roofline']['avg_launch_us'], c2['roofline']['frac'])
print('stock', e['stock_proposal']['value'], e['stock_proposal']['roofline']['frac'], e['stock_proposal']['roofline']['avg_launch_us'])
for k,v in e['kernel_legs'].items(): print(k, v['avg_launch_us'], v['frac'])
PY
