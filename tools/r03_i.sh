#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
start=$(date +%s); python bench.py > $OUT/r03i_bench_default.json 2> $OUT/r03i_bench_default.err; echo "default bench wall: $(( $(date +%s) - start )) s"
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
