#!/bin/bash
# round 3, session A: new tests, the whole GPU suite, c4 bench lines (lazy gather on / off)
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q > $OUT/r03a_round3.txt 2>&1; tail -15 $OUT/r03a_round3.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round3.py > $OUT/r03a_gpu.txt 2>&1; tail -12 $OUT/r03a_gpu.txt
for lazy in 1 0; do
AESMC_LAZY_GATHER=$lazy timeout -k 10 300 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline --extras off > $OUT/r03a_c4_lazy$lazy.json 2> $OUT/r03a_c4_lazy$lazy.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/r03a_c4_lazy$lazy.json").read())
    print("lazy=$lazy", d["value"], d["ms_per_step"], d.get("fwd_bwd_particle_steps_per_sec"), {k: (round(v["avg_us"],1), round(v["GBps"])) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("no json", e); print(open("$OUT/r03a_c4_lazy$lazy.err").read()[-1500:])
PY
done
