#!/bin/bash
# Round 5: the lean K2 walking rows (a workgroup takes several rows in turn; 107 registers, no spills) against one row per
# workgroup — tests under the walk first, then timings.
set -u
OUT=gpurun_out; mkdir -p $OUT
AESMC_K2_ROWS_PER_GROUP=2 timeout -k 10 600 python -m pytest tests/test_gpu_resampler_forms.py tests/test_gpu_kernels.py -m gpu --maxfail=5 -q -x -k "lean or ancestor or resampl" > $OUT/r05k_pytest_walk.txt 2>&1; rc=$?
tail -3 $OUT/r05k_pytest_walk.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/r05k_pytest_walk.txt | head -20 | cut -c1-300; exit $rc; fi
for RPG in 1 2 3 4; do
  AESMC_K2_ROWS_PER_GROUP=$RPG timeout -k 10 200 python tools/k2forms.py 1024,4096 2048,4096 512,4096 256,4096 > $OUT/r05k_k2_rpg$RPG.txt 2>&1; rc=$?
  [ $rc -eq 124 ] && exit $rc
  echo "rows per group $RPG"; grep -E "^B=|rows " $OUT/r05k_k2_rpg$RPG.txt | cut -c1-120
done
