#!/bin/bash
# rocprofv3 kernel traces of the c4 workload with affine callables: forward, and forward + backward.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
prof() {  # name, bench args...
  NAME=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg_prof_$NAME -- \
     python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/lg_prof_$NAME.log 2>&1)
  STATS=$(ls $OUT/lg_prof_$NAME/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 24 > $OUT/lg_rocprof_$NAME.csv
  rm -rf $OUT/lg_prof_$NAME
  cut -c1-150 $OUT/lg_rocprof_$NAME.csv | head -${LINES_SHOWN:-22}
}
W=${WORKLOAD:-c4}
prof fwd_$W --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-backward --extras off
prof fwd_bwd_$W --workload $W --steps 1 --warmup 1 --no-cpu-baseline --extras off
