#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/r03i_gpu.txt 2>&1; tail -6 $OUT/r03i_gpu.txt
for cfg in "c4x2 graph" "c4 graph"; do
set -- $cfg
timeout -k 10 300 python bench.py --workload $1 --mode $2 --steps 5 --warmup 2 --no-cpu-baseline --extras off --no-backward > $OUT/r03i_$1_$2.json 2> $OUT/r03i_$1_$2.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/r03i_$1_$2.json").read())
    print("$1 $2", d["value"], d["ms_per_step"], d.get("mode"), d.get("graph_error"))
except Exception as e:
    print("no json", e); print(open("$OUT/r03i_$1_$2.err").read()[-800:])
PY
done
/usr/bin/time -v python bench.py > $OUT/r03i_bench_default.json 2> $OUT/r03i_bench_default.err; grep "Elapsed\|Maximum resident" $OUT/r03i_bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/r03i_bench_default.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "fwd_bwd_particle_steps_per_sec", "mode")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_us", "achieved", "frac", "traffic", "unique_ancestor_fraction")})
e = d.get("extras", {})
print("projection", e.get("strong_scaling_projection"))
for k in ("stock_proposal", "matmul_callables", "c2_hipgraph", "c4nl"):
    v = e.get(k) or {}
    print(k, v.get("value"), v.get("ms_per_step"), v.get("fwd_bwd_particle_steps_per_sec"), v.get("mode"))
print("cpu", d.get("cpu_baseline"))
print("parity", {k: (v["free_running_rel_dlogZ"], v["teacher_forced_max_rel_dlogw"], v["teacher_forced_flip_rate"]) for k, v in e.get("fp32_fixture_parity", {}).items()})
PY
