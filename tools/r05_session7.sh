#!/bin/bash
# Round 5, seventh GPU session: K14's rows form on the packed pipe — tests, then timings with and without the pairs; the full
# suite; the default bench with rocprofv3 traces.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_fused_step_oracle.py tests/test_gpu_linear_gaussian.py tests/test_gpu_noise_and_lazy_latents.py -m gpu --maxfail=5 -q -x > $OUT/r05g_pytest.txt 2>&1; rc=$?
tail -5 $OUT/r05g_pytest.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/r05g_pytest.txt | head -30 | cut -c1-300; exit $rc; fi
for P in 1 0; do
  AESMC_K14_PAIRS=$P timeout -k 10 300 python tools/k14bench.py > $OUT/r05g_k14_pairs$P.txt 2>&1; rc=$?; stop_if_killed $rc
  echo "pairs $P"; grep -v amdgpu.ids $OUT/r05g_k14_pairs$P.txt | cut -c1-150
done
for D in 4 8 12; do
  timeout -k 10 200 python tools/k14bench.py --dims 1024,4096,$D --only "children (healthy)" > $OUT/r05g_k14_d$D.txt 2>&1; rc=$?; stop_if_killed $rc
  echo "d=$D: $(grep children $OUT/r05g_k14_d$D.txt | cut -c1-140)"
done
bash tools/gpu_session.sh r05g tests
