#!/bin/bash
# usage: tools/quick_bench.sh [workload ...]   — pytest -m gpu, then one short bench line per workload
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for w in "$@"; do
  steps=5; [ "$w" = c4 ] && steps=2
  python bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline 2>gpurun_out/qb_$w.err | python -c '
import sys, json
text = sys.stdin.read()
try:
    d = json.loads(text)
except Exception:
    print("NO JSON", text[-500:]); sys.exit(0)
print(d["config"]["workload"][:3], d["value"], d["ms_per_step"], d["mode"], d["graph_error"], d["fwd_bwd_particle_steps_per_sec"])
print({k: (v["avg_us"], v["GBps"]) for k, v in d["kernels"].items()})'
  tail -3 gpurun_out/qb_$w.err
done
