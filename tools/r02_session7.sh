#!/bin/bash
# What one GPU's shard of the strong-scaling curve costs (B = 1024 / N rows), counters of the fused
# step with the round's final kernel, forward+backward trace at configs[1].
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for W in c4x2 c4x4 c4s; do
  timeout 900 python bench.py --workload $W --steps 5 --warmup 2 --extras off --no-cpu-baseline --no-backward --tunableop-file /tmp/tuned_$W.csv > $OUT/s7_bench_$W.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/s7_bench_$W.json')); r=d['roofline']; print('$W', d['value'], d['ms_per_step'], d['mode'], d['eager_particle_steps_per_sec'], r['avg_launch_us'], r['frac'], r.get('frac_moved_bytes'))"
done
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/s7_pmcstep_$i -- \
     python $GRAFT_REPO_ROOT/tools/pmc_step.py > $OUT/s7_pmcstep_$i.log 2>&1)
  CSV=$(ls $OUT/s7_pmcstep_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/s7_pmcstep_set$i.csv
  rm -rf $OUT/s7_pmcstep_$i
done
python tools/pmc_step_summarize.py $OUT/s7_pmc_step_counters.csv $(ls $OUT/s7_pmcstep_set*.csv)
rm -f $OUT/s7_pmcstep_set*.csv
TUNED2=/tmp/aesmc_tuned_c2.csv
python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $TUNED2 > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s7_profbwd -- \
   python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --extras off \
   --tunableop-file $TUNED2 > $OUT/s7_profbwd.log 2>&1)
STATS=$(ls $OUT/s7_profbwd/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 40 > $OUT/s7_rocprof_bwd_c2_tuned.csv && grep "aesmc::\|^#" $OUT/s7_rocprof_bwd_c2_tuned.csv | cut -c1-160
rm -rf $OUT/s7_profbwd
