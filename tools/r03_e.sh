#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_graphs.py -x -q > $OUT/r03e_round3.txt 2>&1; tail -15 $OUT/r03e_round3.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round3.py --deselect tests/test_gpu_graphs.py > $OUT/r03e_gpu.txt 2>&1; tail -8 $OUT/r03e_gpu.txt
python tools/host_overhead.py --grad 1 > $OUT/r03e_host.txt 2>&1; head -3 $OUT/r03e_host.txt
for w in c4 c2 c4x2 c4x4 c4s; do
timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extras off > $OUT/r03e_$w.json 2> $OUT/r03e_$w.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/r03e_$w.json").read())
    print("$w", d["value"], d["ms_per_step"], d.get("mode"), d.get("fwd_bwd_particle_steps_per_sec"), {k: (round(v["avg_us"],1), round(v["GBps"])) for k, v in d.get("kernels", {}).items() if "propagate" in k or "resample" in k})
except Exception as e:
    print("no json", e); print(open("$OUT/r03e_$w.err").read()[-1500:])
PY
done
