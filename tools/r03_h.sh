#!/bin/bash
set -u
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "noise_inside or whose_kernels or float32_runs" > $OUT/r03h_round3.txt 2>&1; tail -4 $OUT/r03h_round3.txt
for probe in 0 1 2; do
echo "== probe $probe"; AESMC_K16_PROBE=$probe timeout -k 10 300 python tools/k16bench.py 1024 4096 10 2>&1 | grep "K16\|K15 through"
done
