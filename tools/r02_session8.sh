#!/bin/bash
# Final verification of round 2: every GPU test, smoke(), the example, the default bench line, traces.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 600 python examples/lgssm_train.py --steps 40 2>&1 | tail -2
timeout 600 python examples/lgssm_train.py --steps 40 --graph 2>&1 | tail -2
TUNED=/tmp/aesmc_tuned.csv
T0=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 --tunableop-file $TUNED > $OUT/s8_bench_default.json 2> $OUT/s8_bench_default.err
echo "bench default (--steps 20 --warmup 5) wall seconds: $(( $(date +%s) - T0 ))"
python - <<PY
import json
d=json.load(open('$OUT/s8_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras','config','cpu_baseline')}, indent=None)[:1800])
print(json.dumps(d['cpu_baseline'])[:300])
PY
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s8_prof_c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off \
   --tunableop-file $TUNED > $OUT/s8_prof_c4.log 2>&1)
STATS=$(ls $OUT/s8_prof_c4/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 14 > $OUT/s8_rocprof_kernel_stats_c4.csv && head -12 $OUT/s8_rocprof_kernel_stats_c4.csv | cut -c1-150
rm -rf $OUT/s8_prof_c4
TUNED2=/tmp/aesmc_tuned_c2.csv
python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off --tunableop-file $TUNED2 > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s8_prof_c2 -- \
   python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --no-backward --extras off \
   --tunableop-file $TUNED2 > $OUT/s8_prof_c2.log 2>&1)
STATS=$(ls $OUT/s8_prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 14 > $OUT/s8_rocprof_kernel_stats_c2.csv && head -14 $OUT/s8_rocprof_kernel_stats_c2.csv | cut -c1-150
rm -rf $OUT/s8_prof_c2
