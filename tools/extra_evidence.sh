#!/bin/bash
# Evidence beyond tools/gpu_session.sh (run through gpurun from the repository root):   tools/extra_evidence.sh <tag>
# PMC traffic of configs[1] and of the stock-proposal workload, a kernel trace of configs[4]'s shape (c5h), and
# configs[1] as an eager loop.  Everything lands in gpurun_out/<tag>_*.
set -u
TAG=${1:-extra}
OUT=gpurun_out
tools/pmc_traffic.sh c2 tuned 6 2>&1 | tail -6
tools/pmc_traffic.sh c4 stock 6 2>&1 | tail -6
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_c5h -- python $GRAFT_REPO_ROOT/bench.py --workload c5h --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward > $GRAFT_REPO_ROOT/$OUT/${TAG}_c5h.log 2>&1)
STATS=$(ls $OUT/${TAG}_c5h/*/*kernel_stats.csv | head -1)
python tools/summarize_rocprof.py $STATS 16 > $OUT/${TAG}_rocprof_c5h.csv
rm -rf $OUT/${TAG}_c5h
cut -c1-170 $OUT/${TAG}_rocprof_c5h.csv
tail -1 $OUT/${TAG}_c5h.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5h', d['mode'], round(d['ms_per_step'],2), d['value'])"
timeout -k 10 300 python bench.py --workload c2 --mode eager --steps 10 --warmup 3 --extras off --no-cpu-baseline > $OUT/${TAG}_c2_eager.json 2>/dev/null
python -c "
import json; d=json.loads(open('$OUT/${TAG}_c2_eager.json').read().strip().splitlines()[-1]); print('c2 eager', round(d['ms_per_step'],3), d['value'], d.get('fwd_bwd_particle_steps_per_sec'))"
