"""Prints the parts of a bench.py JSON line a session looks at first.    python tools/bench_digest.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "fwd_bwd_particle_steps_per_sec", "mode")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_us", "achieved", "frac", "traffic")})
e = d.get("extras", {})
print("projection", e.get("strong_scaling_projection"))
print("seconds", e.get("bench_seconds"))
for k in ("stock_proposal", "matmul_callables", "c2_hipgraph", "c4nl"):
    v = e.get(k) or {}
    print(k, v.get("value"), v.get("ms_per_step"), v.get("fwd_bwd_particle_steps_per_sec"), v.get("mode"),
          "eager loop:", v.get("eager_particle_steps_per_sec"))
print("cpu", d.get("cpu_baseline"))
