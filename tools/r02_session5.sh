#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/stepbench.py c2 c4s c4 > $OUT/s5_stepbench.txt 2>&1; grep "parts\|==\|K1" $OUT/s5_stepbench.txt
timeout 600 python tools/kbench.py c2 c4 > $OUT/s5_kbench.txt 2>&1; grep -i "backward\|==\|K3 gather" $OUT/s5_kbench.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/s5_pytest_gpu.txt; tail -5 $OUT/s5_pytest_gpu.txt
TUNED=/tmp/aesmc_tuned.csv
/usr/bin/time -v timeout 1500 python bench.py --tunableop-file $TUNED > $OUT/s5_bench_default.json 2> $OUT/s5_bench_default.err
grep -E "Elapsed|Maximum resident" $OUT/s5_bench_default.err
python - <<PY
import json
d=json.load(open('$OUT/s5_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras','config')}, indent=None)[:2500])
for k,v in d.get('extras',{}).items(): print(k, json.dumps(v)[:1200])
PY
# forward + backward trace at c4 (eager, 1 step each)
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s5_profbwd4 -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --extras off \
   --tunableop-file $TUNED > $OUT/s5_profbwd4.log 2>&1)
STATS=$(ls $OUT/s5_profbwd4/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 30 > $OUT/s5_rocprof_fwd_bwd_c4.csv && head -36 $OUT/s5_rocprof_fwd_bwd_c4.csv | cut -c1-180
rm -rf $OUT/s5_profbwd4
