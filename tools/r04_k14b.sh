#!/bin/bash
# gpurun -- tools/r04_k14b.sh : the step backward's launch time only (tools/k14bench.py)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python tools/k14bench.py > $OUT/r04_k14_bench.txt 2>&1 || { tail -20 $OUT/r04_k14_bench.txt; exit 1; }
cat $OUT/r04_k14_bench.txt
