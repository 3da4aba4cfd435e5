#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -q -s -k "float32_runs or whose_kernels" > $OUT/r03f_round3.txt 2>&1; grep "fp32 parity\|passed\|failed\|Error\|assert" $OUT/r03f_round3.txt | head -40
for w in c2 c4s c4x4; do
timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extras off > $OUT/r03f_$w.json 2> $OUT/r03f_$w.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/r03f_$w.json").read())
    print("$w", d["value"], d["ms_per_step"], d.get("mode"), d.get("fwd_bwd_particle_steps_per_sec"), {k: (round(v["avg_us"],1), round(v["GBps"])) for k, v in d.get("kernels", {}).items() if "propagate" in k or "resample" in k or "philox" in k})
except Exception as e:
    print("no json", e); print(open("$OUT/r03f_$w.err").read()[-1500:])
PY
done
