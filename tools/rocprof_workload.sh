#!/bin/bash
# rocprofv3 kernel stats of one bench workload (forward, eager loop and replays, and — unless --no-backward is among the
# extra arguments — the captured forward + backward), condensed into profiles/-ready CSV under gpurun_out/.
#   tools/rocprof_workload.sh WORKLOAD TAG [bench.py arguments...]
# Run on the GPU box (gpurun); `python3 bench.py` itself follows `--` (no wrapper process between rocprofv3 and it).
set -u
WORKLOAD=${1:?workload}
TAG=${2:?tag}
shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OUT/${TAG}_prof_${WORKLOAD} -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WORKLOAD --steps 5 --warmup 1 \
    --no-cpu-baseline --extras off "$@" > $OUT/${TAG}_prof_${WORKLOAD}.json 2> $OUT/${TAG}_prof_${WORKLOAD}.err)
STATS=$(ls $OUT/${TAG}_prof_${WORKLOAD}/*/*kernel_stats.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $STATS 40 > $OUT/${TAG}_rocprof_${WORKLOAD}.csv
rm -rf $OUT/${TAG}_prof_${WORKLOAD}
head -45 $OUT/${TAG}_rocprof_${WORKLOAD}.csv | cut -c1-200
