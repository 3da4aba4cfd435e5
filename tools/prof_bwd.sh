#!/bin/bash
# rocprof kernel stats of the forward+backward ELBO (graph-captured training step) at c2
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OUT/${TAG}_profbwd -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 \
    --no-cpu-baseline --tunableop ${TUNABLE:-off} > $OUT/${TAG}_profbwd.log 2>&1)
STATS=$(ls $OUT/${TAG}_profbwd/*/*kernel_stats.csv | head -1)
python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $STATS 40 > $OUT/${TAG}_rocprof_bwd_c2.csv
rm -rf $OUT/${TAG}_profbwd
head -50 $OUT/${TAG}_rocprof_bwd_c2.csv | cut -c1-220
