#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r03l_prof -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --extras off > $GRAFT_REPO_ROOT/$OUT/r03l_prof.log 2>&1)
STATS=$(ls $OUT/r03l_prof/*/*kernel_stats.csv | head -1)
python tools/summarize_rocprof.py $STATS 24 > $OUT/r03l_rocprof_fwd_bwd_c4.csv
rm -rf $OUT/r03l_prof
cut -c1-200 $OUT/r03l_rocprof_fwd_bwd_c4.csv | head -34
start=$(date +%s); python bench.py > $OUT/r03l_bench_default.json 2> $OUT/r03l_bench_default.err; echo "default bench wall: $(( $(date +%s) - start )) s"
python - <<PY
import json
d = json.loads(open("$OUT/r03l_bench_default.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "fwd_bwd_particle_steps_per_sec", "mode")})
e = d.get("extras", {})
print("projection", e.get("strong_scaling_projection"))
print("seconds", e.get("bench_seconds"))
for k in ("stock_proposal", "matmul_callables", "c2_hipgraph", "c4nl"):
    v = e.get(k) or {}
    print(k, v.get("value"), v.get("ms_per_step"), v.get("fwd_bwd_particle_steps_per_sec"), v.get("mode"))
PY
