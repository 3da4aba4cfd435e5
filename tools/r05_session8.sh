#!/bin/bash
# Round 5, eighth GPU session: the differentiable wide step; the renamed suites; the settings object in the GPU paths.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_fused_step_oracle.py tests/test_gpu_configs_parity.py tests/test_gpu_graphs.py tests/test_gpu_infer.py tests/test_gpu_reference_suite.py -m gpu --maxfail=6 -q > $OUT/r05h_pytest.txt 2>&1; rc=$?
tail -5 $OUT/r05h_pytest.txt | cut -c1-300
[ $rc -ne 0 ] && grep -n "Error\|assert\|FAILED" $OUT/r05h_pytest.txt | head -40 | cut -c1-300
exit $rc
