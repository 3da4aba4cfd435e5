#!/bin/bash
# gpurun -- tools/r04_c5h.sh : configs[4]'s shape (c5h): the bench line and a kernel trace
set -u
OUT=gpurun_out; mkdir -p $OUT; TAG=r04
timeout -k 10 600 python bench.py --workload c5h --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward > $OUT/${TAG}_bench_c5h.json 2> $OUT/${TAG}_bench_c5h.err || { tail -20 $OUT/${TAG}_bench_c5h.err; exit 1; }
tail -1 $OUT/${TAG}_bench_c5h.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5h', d['mode'], round(d['ms_per_step'],2), d['value'])"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_c5h -- python $GRAFT_REPO_ROOT/bench.py --workload c5h --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward > $GRAFT_REPO_ROOT/$OUT/${TAG}_c5h.log 2>&1) || { tail -20 $OUT/${TAG}_c5h.log; exit 1; }
STATS=$(ls $OUT/${TAG}_c5h/*/*kernel_stats.csv | head -1)
python tools/summarize_rocprof.py $STATS 16 > $OUT/${TAG}_rocprof_kernel_stats_c5h.csv
rm -rf $OUT/${TAG}_c5h
cut -c1-170 $OUT/${TAG}_rocprof_kernel_stats_c5h.csv
