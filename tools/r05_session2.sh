#!/bin/bash
# Round 5, second GPU session: the item form of K16 — bit-equality with the persistent form and the C oracle, then the
# two forms side by side at the strong-scaling shard sizes, configs[1]'s shape and other extents.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_propagation_forms.py tests/test_gpu_fused_step_oracle.py -m gpu --maxfail=5 -q -x > $OUT/r05b_pytest_forms.txt 2>&1; rc=$?
tail -5 $OUT/r05b_pytest_forms.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/r05b_pytest_forms.txt | head -30 | cut -c1-300; exit $rc; fi
timeout -k 10 600 python tools/k16forms.py 128,4096,10 256,4096,10 512,4096,10 1024,4096,10 256,1024,10 1024,4096,4 1024,4096,8 1024,4096,12 128,4096,8 > $OUT/r05b_k16forms.txt 2>&1; rc=$?
grep -v amdgpu.ids $OUT/r05b_k16forms.txt | cut -c1-160
exit $rc
