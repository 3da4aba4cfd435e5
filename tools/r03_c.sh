#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "noise_inside or philox" > $OUT/r03c_round3.txt 2>&1; tail -5 $OUT/r03c_round3.txt
timeout -k 10 300 python tools/k16bench.py 1024 4096 10 > $OUT/r03c_k16_c4.txt 2>&1; cat $OUT/r03c_k16_c4.txt
bash tools/pmc_k16.sh > $OUT/r03c_pmc.log 2>&1; tail -12 $OUT/r03c_pmc.log
