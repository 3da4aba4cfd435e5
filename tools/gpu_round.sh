#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, benches, rocprof kernel traces.
# Usage: tools/gpu_round.sh <tag> [workloads...]
set -u
TAG=${1:-r01}; shift || true
WORKLOADS=${@:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.txt
tail -3 $OUT/${TAG}_pytest_gpu.txt
for W in $WORKLOADS; do
  STEPS=5; [ "$W" != "c2" ] && STEPS=2
  TUNED=/tmp/aesmc_tuned_$W.csv
  python bench.py --workload $W --steps $STEPS --warmup 1 --tunableop-file $TUNED > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
  cat $OUT/${TAG}_bench_$W.json
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv \
      -d $OUT/${TAG}_prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 \
      --no-cpu-baseline --no-backward --tunableop-file $TUNED > $OUT/${TAG}_prof_$W.log 2>&1)
  STATS=$(ls $OUT/${TAG}_prof_$W/*/*kernel_stats.csv | head -1)
  python tools/summarize_rocprof.py $STATS 14 > $OUT/${TAG}_rocprof_$W.csv
  rm -rf $OUT/${TAG}_prof_$W
  head -30 $OUT/${TAG}_rocprof_$W.csv
done
