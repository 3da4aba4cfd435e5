#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q > $OUT/r03d_round3.txt 2>&1; tail -15 $OUT/r03d_round3.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round3.py > $OUT/r03d_gpu.txt 2>&1; tail -8 $OUT/r03d_gpu.txt
for cfg in "1 1" "1 0" "0 0"; do
set -- $cfg
AESMC_LAZY_GATHER=$1 AESMC_KERNEL_NOISE=$2 timeout -k 10 300 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline --extras off > $OUT/r03d_c4_$1$2.json 2> $OUT/r03d_c4_$1$2.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/r03d_c4_$1$2.json").read())
    print("lazy=$1 noise=$2", d["value"], d["ms_per_step"], d.get("fwd_bwd_particle_steps_per_sec"), {k: (round(v["avg_us"],1), round(v["GBps"])) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("no json", e); print(open("$OUT/r03d_c4_$1$2.err").read()[-1500:])
PY
done
