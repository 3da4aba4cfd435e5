#!/bin/bash
# Round 5, fifth GPU session: the lean K2's register budget (one round of workgroups at 64 registers against two at 76); K14's
# rows form at the other even extents — bit-equality with the tiles form first, then both forms' timings.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
for MW in 8; do
  AESMC_K2_MIN_WAVES=$MW timeout -k 10 200 python tools/k2forms.py 1024,4096 512,4096 128,4096 256,1024 > $OUT/r05e_k2_minw$MW.txt 2>&1; rc=$?; stop_if_killed $rc
  echo "min waves $MW"; grep -E "^B=|rows " $OUT/r05e_k2_minw$MW.txt | cut -c1-120
done
timeout -k 10 900 python -m pytest tests/test_gpu_fused_step_oracle.py tests/test_gpu_resampler_forms.py -m gpu --maxfail=5 -q -x -k "step_backward or lean" > $OUT/r05e_pytest.txt 2>&1; rc=$?
tail -5 $OUT/r05e_pytest.txt | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/r05e_pytest.txt | head -30 | cut -c1-300; exit $rc; fi
for D in 4 8 12 14 10; do
  for FORM in rows tiles; do
    AESMC_K14_FORM=$FORM timeout -k 10 200 python tools/k14bench.py --dims 1024,4096,$D --only "children (healthy)" > $OUT/r05e_k14_d${D}_$FORM.txt 2>&1; rc=$?; stop_if_killed $rc
    echo "d=$D $FORM: $(grep children $OUT/r05e_k14_d${D}_$FORM.txt | cut -c1-140)"
  done
done
timeout -k 10 900 python -m pytest tests -m gpu --maxfail=8 -q > $OUT/r05e_pytest_all.txt 2>&1; rc=$?
tail -4 $OUT/r05e_pytest_all.txt | cut -c1-300
[ $rc -ne 0 ] && grep -n "Error\|assert\|FAILED" $OUT/r05e_pytest_all.txt | head -40 | cut -c1-300
stop_if_killed $rc
timeout -k 10 200 python tools/host_overhead.py --grad 1 > $OUT/r05e_host_overhead_grad.txt 2>&1; rc=$?; stop_if_killed $rc; grep "host us" $OUT/r05e_host_overhead_grad.txt
timeout -k 10 200 python tools/host_overhead.py --grad 1 --affine 0 > $OUT/r05e_host_overhead_matmul.txt 2>&1; rc=$?; stop_if_killed $rc; grep "host us" $OUT/r05e_host_overhead_matmul.txt
timeout -k 10 600 python bench.py > $OUT/r05e_bench_default.json 2> $OUT/r05e_bench_default.err; rc=$?; stop_if_killed $rc
python tools/bench_digest.py $OUT/r05e_bench_default.json
