#!/bin/bash
# One GPU session (run through gpurun from the repository root):
#   tools/gpu_session.sh <tag> [tests|notests] [pytest -k expression]
# 1. the -m gpu parity tests (or the subset named), 2. the default bench line, 3. rocprofv3 kernel traces of the
# north-star workload (forward + backward, and forward only).  Everything lands in gpurun_out/<tag>_*.
set -u
TAG=${1:-session}; TESTS=${2:-tests}; EXPR=${3:-}
OUT=gpurun_out
mkdir -p $OUT
if [ "$TESTS" = "tests" ]; then
  if [ -n "$EXPR" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu --maxfail=8 -q -k "$EXPR" > $OUT/${TAG}_pytest_gpu.txt 2>&1
  else
    timeout -k 10 900 python -m pytest tests -m gpu --maxfail=8 -q > $OUT/${TAG}_pytest_gpu.txt 2>&1
  fi
  rc=$?
  tail -5 $OUT/${TAG}_pytest_gpu.txt | cut -c1-300
  if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/${TAG}_pytest_gpu.txt | head -40 | cut -c1-300; exit $rc; fi
fi
start=$(date +%s)
timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err || { tail -20 $OUT/${TAG}_bench_default.err; exit 1; }
echo "default bench wall: $(( $(date +%s) - start )) s"
python tools/bench_digest.py $OUT/${TAG}_bench_default.json
for LEG in fwd_bwd fwd; do
  EXTRA=""; [ "$LEG" = "fwd" ] && EXTRA="--no-backward"
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv \
      -d $GRAFT_REPO_ROOT/$OUT/${TAG}_prof_$LEG -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 2 --warmup 1 \
      --no-cpu-baseline --extras off $EXTRA > $GRAFT_REPO_ROOT/$OUT/${TAG}_prof_$LEG.log 2>&1) || { tail -20 $OUT/${TAG}_prof_$LEG.log; exit 1; }
  STATS=$(ls $OUT/${TAG}_prof_$LEG/*/*kernel_stats.csv | head -1)
  python tools/summarize_rocprof.py $STATS 24 > $OUT/${TAG}_rocprof_${LEG}_c4.csv
  rm -rf $OUT/${TAG}_prof_$LEG
  cut -c1-160 $OUT/${TAG}_rocprof_${LEG}_c4.csv | head -14
done
