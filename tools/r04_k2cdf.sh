#!/bin/bash
# gpurun -- tools/r04_k2cdf.sh : the float32-CDF experiment (tests + timing)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -s -k "float32_cdf" > $OUT/r04_k2cdf_tests.txt 2>&1
rc=$?
grep "float32 cdf\|passed\|failed" $OUT/r04_k2cdf_tests.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED\|Mismatch" $OUT/r04_k2cdf_tests.txt | head -40 | cut -c1-300; exit $rc; fi
timeout -k 10 300 python tools/k2cdfbench.py c2 c4 c5 > $OUT/r04_k2cdf_bench.txt 2>&1 || { tail -20 $OUT/r04_k2cdf_bench.txt; exit 1; }
cat $OUT/r04_k2cdf_bench.txt
