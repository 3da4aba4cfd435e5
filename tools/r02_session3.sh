#!/bin/bash
# GPU session 3 of round 2: all GPU tests, step variants (parts / preload), backward kernels A/B,
# the default bench line, forward+backward trace at configs[1].
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/s3_pytest_gpu.txt
tail -12 $OUT/s3_pytest_gpu.txt
timeout 600 python tools/stepbench.py c2 c4s c4 > $OUT/s3_stepbench.txt 2>&1
grep "preload\|==" $OUT/s3_stepbench.txt
timeout 600 python tools/kbench.py c2 c4 > $OUT/s3_kbench.txt 2>&1
grep -i "backward\|==" $OUT/s3_kbench.txt
TUNED=/tmp/aesmc_tuned.csv
timeout 1500 python bench.py --tunableop-file $TUNED > $OUT/s3_bench_default.json 2> $OUT/s3_bench_default.err
tail -3 $OUT/s3_bench_default.err
python - <<PY
import json
d=json.load(open('$OUT/s3_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras','config')}, indent=None))
for k,v in d.get('extras',{}).items(): print(k, json.dumps(v)[:3000])
PY
# forward + backward at configs[1], graph-captured, TunableOp picks cached by a first run
TUNED2=/tmp/aesmc_tuned_c2.csv
python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --extras off --tunableop-file $TUNED2 > $OUT/s3_bench_c2.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/s3_bench_c2.json')); print('c2', d['value'], d['ms_per_step'], d['fwd_bwd_particle_steps_per_sec'], d['roofline'])"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s3_profbwd -- \
   python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --extras off \
   --tunableop-file $TUNED2 > $OUT/s3_profbwd.log 2>&1)
STATS=$(ls $OUT/s3_profbwd/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 40 > $OUT/s3_rocprof_bwd_c2_tuned.csv && head -45 $OUT/s3_rocprof_bwd_c2_tuned.csv | cut -c1-200
rm -rf $OUT/s3_profbwd
