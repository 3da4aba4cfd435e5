#!/bin/bash
# Round 5, fourth GPU session: rows per workgroup of the lean K2; the fused launch against fill + K15 at small shapes (where
# DRAWN_MIN_PARTICLES should sit); then the full -m gpu suite, the default bench line and its rocprofv3 traces.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
for RPG in 1 2 4; do
  AESMC_K2_ROWS_PER_GROUP=$RPG timeout -k 10 200 python tools/k2forms.py 1024,4096 2048,4096 512,4096 > $OUT/r05d_k2_rpg$RPG.txt 2>&1; rc=$?; stop_if_killed $rc
  echo "rows per group $RPG"; grep -E "^B=|rows " $OUT/r05d_k2_rpg$RPG.txt | cut -c1-120
done
timeout -k 10 400 python tools/k16forms.py 64,1024,10 16,1024,10 256,256,10 8,128,10 32,4096,10 64,512,10 > $OUT/r05d_k16_small.txt 2>&1; rc=$?; stop_if_killed $rc
grep -v amdgpu.ids $OUT/r05d_k16_small.txt | cut -c1-140
bash tools/gpu_session.sh r05d tests
