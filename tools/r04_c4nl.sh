#!/bin/bash
# configs[3] (nonlinear SSM + MLP proposal) on one GPU's shard: kernel traces of the forward and of forward + backward
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LEG in fwd fwd_bwd; do
  EXTRA=""; [ "$LEG" = "fwd" ] && EXTRA="--no-backward"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_c4nl_$LEG -- python $GRAFT_REPO_ROOT/bench.py --workload c4nl --steps 2 --warmup 1 --no-cpu-baseline --extras off --mode eager $EXTRA > $OUT/r04_c4nl_$LEG.log 2>&1 || { tail -20 $OUT/r04_c4nl_$LEG.log; exit 1; }
  STATS=$(ls $OUT/r04_c4nl_$LEG/*/*kernel_stats.csv | head -1)
  python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $STATS 30 > $OUT/r04_rocprof_c4nl_$LEG.csv
  rm -rf $OUT/r04_c4nl_$LEG
  cut -c1-170 $OUT/r04_rocprof_c4nl_$LEG.csv | head -34
  tail -1 $OUT/r04_c4nl_$LEG.log | cut -c1-300
done
