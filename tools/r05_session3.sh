#!/bin/bash
# Round 5, third GPU session: the lean K2 and the item form of K16 at every extent — tests first, then timings.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_resampler_forms.py tests/test_gpu_propagation_forms.py tests/test_gpu_kernels.py -m gpu --maxfail=5 -q -x > $OUT/r05c_pytest.txt 2>&1; rc=$?
tail -5 $OUT/r05c_pytest.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/r05c_pytest.txt | head -30 | cut -c1-300; exit $rc; fi
timeout -k 10 300 python tools/k2forms.py > $OUT/r05c_k2forms.txt 2>&1; rc=$?
grep -v amdgpu.ids $OUT/r05c_k2forms.txt | cut -c1-160
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/k16forms.py 1024,4096,16 1024,4096,14 1024,4096,5 1024,4096,3 1024,4096,2 128,4096,16 > $OUT/r05c_k16forms.txt 2>&1; rc=$?
grep -v amdgpu.ids $OUT/r05c_k16forms.txt | cut -c1-160
exit $rc
