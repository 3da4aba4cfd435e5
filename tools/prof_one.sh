#!/bin/bash
# rocprofv3 kernel trace of one bench workload (forward): tools/prof_one.sh <workload> [extra bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
W=$1; shift
cd $GRAFT_REPO_ROOT
T=/tmp/aesmc_tuned_$W.csv
python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --extras off --no-backward --tunableop-file $T "$@" > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_prof_$W -- \
   python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward --tunableop-file $T "$@" > $OUT/one_prof_$W.log 2>&1)
STATS=$(ls $OUT/one_prof_$W/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$STATS" ] && python tools/summarize_rocprof.py $STATS 16 > $OUT/one_rocprof_$W.csv
rm -rf $OUT/one_prof_$W
cut -c1-140 $OUT/one_rocprof_$W.csv | head -22
grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*' $OUT/one_prof_$W.log | head -2
