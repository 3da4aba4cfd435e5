#!/bin/bash
# gpurun -- tools/r04_k14.sh : both forms of the step's backward bit for bit, the existing K14 tests, then the launch's time
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_fused_step_oracle.py tests/test_gpu_noise_and_lazy_latents.py tests/test_gpu_linear_gaussian.py -m gpu -q -x -k "step_backward or both_forms_of_the_step" > $OUT/r04_k14_tests.txt 2>&1
rc=$?
tail -3 $OUT/r04_k14_tests.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED\|Mismatch" $OUT/r04_k14_tests.txt | head -40 | cut -c1-300; exit $rc; fi
timeout -k 10 300 python tools/k14bench.py > $OUT/r04_k14_bench.txt 2>&1 || { tail -20 $OUT/r04_k14_bench.txt; exit 1; }
cat $OUT/r04_k14_bench.txt
AESMC_K14_FORM=tiles timeout -k 10 300 python tools/k14bench.py > $OUT/r04_k14_bench_tiles.txt 2>&1 || { tail -20 $OUT/r04_k14_bench_tiles.txt; exit 1; }
cat $OUT/r04_k14_bench_tiles.txt
