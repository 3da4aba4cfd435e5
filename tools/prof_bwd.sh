#!/bin/bash
# rocprof kernel stats of the forward+backward ELBO (graph-captured training step) at c2.
# TUNABLE=on: a first, unprofiled run leaves TunableOp's picks in a file, so the profiled run holds
# the steady state only (no tuning trials).
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
TUNED=/tmp/aesmc_tuned_bwd.csv
EXTRA="--tunableop ${TUNABLE:-off}"
if [ "${TUNABLE:-off}" = "on" ]; then
  python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --tunableop-file $TUNED > /dev/null 2>&1
  EXTRA="--tunableop on --tunableop-file $TUNED"
fi
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OUT/${TAG}_profbwd -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 \
    --no-cpu-baseline $EXTRA > $OUT/${TAG}_profbwd.log 2>&1)
STATS=$(ls $OUT/${TAG}_profbwd/*/*kernel_stats.csv | head -1)
python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $STATS 40 > $OUT/${TAG}_rocprof_bwd_c2.csv
rm -rf $OUT/${TAG}_profbwd
head -50 $OUT/${TAG}_rocprof_bwd_c2.csv | cut -c1-220
