#!/bin/bash
# Where does a whole-ELBO hipGraph at B=512 K=4096 spend its time?  Kernel trace of graph mode beside eager mode.
set -u
OUT=gpurun_out
for MODE in graph eager; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r03t_$MODE -- \
     python $GRAFT_REPO_ROOT/bench.py --workload c4x2 --mode $MODE --steps 8 --warmup 2 --extras off --no-cpu-baseline --no-backward > $GRAFT_REPO_ROOT/$OUT/r03t_$MODE.log 2>&1)
  STATS=$(ls $OUT/r03t_$MODE/*/*kernel_stats.csv | head -1)
  python tools/summarize_rocprof.py $STATS 8 > $OUT/r03t_rocprof_c4x2_$MODE.csv
  TRACE=$(ls $OUT/r03t_$MODE/*/*kernel_trace.csv | head -1)
  python - <<PY
import csv
rows = list(csv.DictReader(open("$TRACE")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k16 = [r for r in rows if "affine_propagate_noise" in r["Kernel_Name"] or "affine_logweight_kernel" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k16]
n = len(dur)
print("$MODE: propagation launches", n, "first-quarter avg %.1f us, last-quarter avg %.1f us" % (sum(dur[:n//4]) / (n//4), sum(dur[-(n//4):]) / (n//4)))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
print("$MODE: trace span %.1f ms, kernels busy %.1f ms" % (span, busy))
PY
  rm -rf $OUT/r03t_$MODE
  tail -1 $OUT/r03t_$MODE.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE', d['mode'], round(d['ms_per_step'],3))"
  cut -c1-150 $OUT/r03t_rocprof_c4x2_$MODE.csv | head -8
done
