#!/bin/bash
# round 4: the fused (matrix-core) form of K16 — its parity tests, then the step's launches timed in both forms
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_noise_and_lazy_latents.py tests/test_gpu_fused_step_oracle.py -m gpu -q -x -k "noise_inside or c_oracle" > $OUT/r04_k16_tests.txt 2>&1
rc=$?; tail -15 $OUT/r04_k16_tests.txt | cut -c1-250
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/k16bench.py 1024 4096 10 > $OUT/r04_k16bench_fused.txt 2>&1 && tail -4 $OUT/r04_k16bench_fused.txt
AESMC_K16_FORM=roles timeout -k 10 300 python tools/k16bench.py 1024 4096 10 > $OUT/r04_k16bench_roles.txt 2>&1 && tail -2 $OUT/r04_k16bench_roles.txt
