#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/bwdprobe.py > $OUT/s4_bwdprobe.txt 2>&1; cat $OUT/s4_bwdprobe.txt
timeout 600 python tools/stepbench.py c2 c4s c4 > $OUT/s4_stepbench.txt 2>&1; grep "parts\|==\|K1" $OUT/s4_stepbench.txt
timeout 300 python tools/k7bench.py > $OUT/s4_k7bench.txt 2>&1; cat $OUT/s4_k7bench.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/s4_pytest_gpu.txt; tail -5 $OUT/s4_pytest_gpu.txt
for W in c5 c5h; do
  timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --extras off --tunableop-file /tmp/tuned_$W.csv > $OUT/s4_bench_$W.json 2> $OUT/s4_bench_$W.err
  python -c "
import json; d=json.load(open('$OUT/s4_bench_$W.json')); print('$W', d['value'], d['ms_per_step'], json.dumps(d['roofline']), d['cpu_baseline']['value'], d['peak_memory_GB'])" || tail -5 $OUT/s4_bench_$W.err
done
timeout 900 python bench.py --workload c4 --proposal stock --steps 2 --warmup 1 --extras off --no-cpu-baseline --tunableop-file /tmp/tuned_c4.csv > $OUT/s4_bench_c4_stock.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/s4_bench_c4_stock.json')); print('c4 stock fwd_bwd', d['value'], d['fwd_bwd_particle_steps_per_sec'])"
