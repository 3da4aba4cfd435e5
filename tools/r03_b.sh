#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "noise_inside" > $OUT/r03b_round3.txt 2>&1; tail -25 $OUT/r03b_round3.txt
timeout -k 10 300 python tools/k16bench.py 1024 4096 10 > $OUT/r03b_k16_c4.txt 2>&1; cat $OUT/r03b_k16_c4.txt
timeout -k 10 300 python tools/k16bench.py 256 1024 10 > $OUT/r03b_k16_c2.txt 2>&1; cat $OUT/r03b_k16_c2.txt
timeout -k 10 300 python tools/k16bench.py 128 4096 10 > $OUT/r03b_k16_c4s.txt 2>&1; cat $OUT/r03b_k16_c4s.txt
