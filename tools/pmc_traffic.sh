#!/bin/bash
# HBM bytes per launch of the path's kernels on the bench workload's own operands:
#   tools/pmc_traffic.sh <workload> <proposal> [timesteps]
# two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: one counter per run, kernel trace only) over tools/pmc_workload.py,
# merged into gpurun_out/pmc_traffic.json (started from profiles/pmc_traffic.json), which bench.py reads from profiles/.
set -u
W=${1:-c4}; P=${2:-tuned}; STEPS=${3:-6}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
[ -f $OUT/pmc_traffic.json ] || cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmcw_$C -- \
     python $GRAFT_REPO_ROOT/tools/pmc_workload.py $W $P $STEPS > $OUT/pmcw_$C.log 2>&1) || { tail -5 $OUT/pmcw_$C.log; exit 1; }
  CSV=$(ls $OUT/pmcw_$C/*/*counter_collection.csv 2>/dev/null | head -1)
  cp $CSV $OUT/pmcw_$C.csv
  rm -rf $OUT/pmcw_$C
  tail -1 $OUT/pmcw_$C.log
done
python tools/pmc_workload_summarize.py $W $P $OUT/pmcw_FETCH_SIZE.csv $OUT/pmcw_WRITE_SIZE.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic_${W}_${P}.txt
python - <<PY
import json
t = json.load(open("$OUT/pmc_traffic.json"))["$W:$P"]
for k, v in sorted(t.items()):
    if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
        print("%-34s %8.1f MB per launch (fetch %.1f, write %.1f; algorithmic %s) over %d launches" % (
            k, v["hbm_bytes_per_launch"] / 1e6, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6,
            ("%.1f" % (v["algorithmic_bytes_per_launch"] / 1e6)) if "algorithmic_bytes_per_launch" in v else "-", v["launches"]))
PY
rm -f $OUT/pmcw_*.csv $OUT/pmcw_*.log
