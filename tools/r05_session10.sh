#!/bin/bash
# Round 5: the tree as it stands — smoke(), the full -m gpu suite, the default bench line.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05j_smoke.txt 2>&1; rc=$?; tail -2 $OUT/r05j_smoke.txt | cut -c1-400; stop_if_killed $rc
timeout -k 10 900 python -m pytest tests -m gpu --maxfail=8 -q > $OUT/r05j_pytest_gpu.txt 2>&1; rc=$?
tail -4 $OUT/r05j_pytest_gpu.txt | cut -c1-300
[ $rc -ne 0 ] && grep -n "Error\|assert\|FAILED" $OUT/r05j_pytest_gpu.txt | head -40 | cut -c1-300
stop_if_killed $rc
timeout -k 10 600 python bench.py > $OUT/r05j_bench_default.json 2> $OUT/r05j_bench_default.err; rc=$?; stop_if_killed $rc
python tools/bench_digest.py $OUT/r05j_bench_default.json | cut -c1-400
