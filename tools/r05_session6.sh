#!/bin/bash
# Round 5, sixth GPU session: packed multiply-adds in the item form of K16 — bit-equality tests, then timings beside the
# v_fmac chains; the tests that failed on the round's new defaults.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_propagation_forms.py tests/test_gpu_fused_step_oracle.py tests/test_gpu_noise_and_lazy_latents.py tests/test_gpu_graphs.py tests/test_gpu_infer.py -m gpu --maxfail=5 -q > $OUT/r05f_pytest.txt 2>&1; rc=$?
tail -5 $OUT/r05f_pytest.txt | cut -c1-300
[ $rc -ne 0 ] && grep -n "Error\|assert\|FAILED" $OUT/r05f_pytest.txt | head -30 | cut -c1-300
stop_if_killed $rc
timeout -k 10 600 python tools/k16forms.py 1024,4096,10 128,4096,10 256,4096,10 256,1024,10 1024,4096,4 1024,4096,8 1024,4096,12 1024,4096,16 > $OUT/r05f_k16forms.txt 2>&1; rc=$?
grep -v amdgpu.ids $OUT/r05f_k16forms.txt | grep -v "noise fill\|K15" | cut -c1-160
exit $rc
