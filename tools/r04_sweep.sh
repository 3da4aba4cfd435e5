#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r04_k16bench_sweep.txt
for shape in "512 4096 10" "256 4096 10" "128 4096 10" "256 1024 10" "1024 4096 4" "1024 4096 8" "1024 4096 16" "64 16384 10"; do
  AESMC_K16_MIN_PARTICLES=0 timeout -k 10 200 python tools/k16bench.py $shape 2>&1 | grep -v amdgpu.ids | grep "B=\|K2 anc\|philox fill\|K15 through\|K16" >> $OUT/r04_k16bench_sweep.txt
done
cat $OUT/r04_k16bench_sweep.txt | cut -c1-110
