#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_k16prof -- python $GRAFT_REPO_ROOT/tools/k16bench.py 1024 4096 10 > $OUT/r04_k16prof.log 2>&1
STATS=$(ls $OUT/r04_k16prof/*/*kernel_stats.csv | head -1)
python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $STATS 14 | cut -c1-200
rm -rf $OUT/r04_k16prof
