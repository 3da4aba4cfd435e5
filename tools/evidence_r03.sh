set -u
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "falsely or genealogy or graph or train" > $OUT/r03x_tests.txt 2>&1; tail -3 $OUT/r03x_tests.txt
tools/pmc_traffic.sh c2 tuned 6 2>&1 | tail -6
tools/pmc_traffic.sh c4 stock 6 2>&1 | tail -6
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r03x_c5h -- python $GRAFT_REPO_ROOT/bench.py --workload c5h --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward > $GRAFT_REPO_ROOT/$OUT/r03x_c5h.log 2>&1)
STATS=$(ls $OUT/r03x_c5h/*/*kernel_stats.csv | head -1)
python tools/summarize_rocprof.py $STATS 16 > $OUT/r03x_rocprof_c5h.csv
rm -rf $OUT/r03x_c5h
cut -c1-170 $OUT/r03x_rocprof_c5h.csv
tail -1 $OUT/r03x_c5h.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5h', d['mode'], round(d['ms_per_step'],2), d['value'])"
timeout -k 10 300 python bench.py --workload c2 --mode eager --steps 10 --warmup 3 --extras off --no-cpu-baseline > $OUT/r03x_c2_eager.json 2>/dev/null
python -c "
import json; d=json.loads(open('$OUT/r03x_c2_eager.json').read().strip().splitlines()[-1]); print('c2 eager', round(d['ms_per_step'],3), d['value'], d.get('fwd_bwd_particle_steps_per_sec'))"
