#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/kbench.py c2 c4 > $OUT/s6_kbench.txt 2>&1; grep -i "K5\|==" $OUT/s6_kbench.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
TUNED=/tmp/aesmc_tuned.csv
T0=$(date +%s)
timeout 1500 python bench.py --tunableop-file $TUNED > $OUT/s6_bench_default.json 2> $OUT/s6_bench_default.err
echo "bench default wall seconds: $(( $(date +%s) - T0 ))"
python - <<PY
import json
d=json.load(open('$OUT/s6_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras','config')}, indent=None)[:2500])
for k,v in d.get('extras',{}).items(): print(k, json.dumps(v)[:900])
PY
T0=$(date +%s)
timeout 900 python bench.py --tunableop-file $TUNED --extras off --no-cpu-baseline > $OUT/s6_bench_default_cached.json 2>/dev/null
echo "bench (cached tuning, no extras) wall seconds: $(( $(date +%s) - T0 ))"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s6_profbwd4 -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --extras off \
   --tunableop-file $TUNED > $OUT/s6_profbwd4.log 2>&1)
STATS=$(ls $OUT/s6_profbwd4/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 24 > $OUT/s6_rocprof_fwd_bwd_c4.csv && head -40 $OUT/s6_rocprof_fwd_bwd_c4.csv | cut -c1-170
rm -rf $OUT/s6_profbwd4
