#!/bin/bash
# Kernel trace of one workload's forward + backward (default: configs[3]'s nonlinear model on a one-GPU shard):
#   tools/shard_probe.sh [workload] [extra bench.py flags]
set -u
W=${1:-c4nl}; shift || true
OUT=gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/probe_$W -- \
   python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --extras off "$@" > $GRAFT_REPO_ROOT/$OUT/probe_$W.log 2>&1)
STATS=$(ls $OUT/probe_$W/*/*kernel_stats.csv | head -1)
python tools/summarize_rocprof.py $STATS 22 > $OUT/probe_rocprof_$W.csv
rm -rf $OUT/probe_$W
cut -c1-150 $OUT/probe_rocprof_$W.csv
tail -1 $OUT/probe_$W.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['mode'], round(d['ms_per_step'],2), d['value'], d.get('fwd_bwd_particle_steps_per_sec'))"
