#!/bin/bash
# Round 5, ninth GPU session: configs[4]'s shape WITH gradients (the recomputing wide backward), forward + backward timed.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python bench.py --workload c5h --grad on --steps 2 --warmup 1 --extras off --no-cpu-baseline > $OUT/r05i_bench_c5h_grad.json 2> $OUT/r05i_bench_c5h_grad.err; rc=$?
tail -3 $OUT/r05i_bench_c5h_grad.err | cut -c1-300
python - <<PY
import json
try:
    d = json.load(open("$OUT/r05i_bench_c5h_grad.json"))
    print("c5h grad on:", round(d["ms_per_step"], 1), "ms fwd;", "fwd+bwd", d.get("fwd_bwd_particle_steps_per_sec"), d.get("fwd_bwd_error"), "peak GB", d["peak_memory_GB"], "mode", d["mode"])
    print({k: (round(v["avg_us"], 1), v["launches"]) for k, v in d["kernels"].items()})
except Exception as e:
    print("no line", e)
PY
exit $rc
