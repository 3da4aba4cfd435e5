#!/bin/bash
# Launch-by-launch timeline of the north-star workload's forward + backward (start, duration, gap to the launch before):
#   tools/timeline.sh [workload] [first row] [rows]
set -u
W=${1:-c4}; FIRST=${2:-0}; ROWS=${3:-60}
OUT=gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tl_$W -- \
   python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --extras off > $GRAFT_REPO_ROOT/$OUT/tl_$W.log 2>&1)
TRACE=$(ls $OUT/tl_$W/*/*kernel_trace.csv | head -1)
python - "$TRACE" "$FIRST" "$ROWS" > $OUT/timeline_$W.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last launch of the step's backward kernel marks the end of the last backward pass: print what leads up to it
last = max(i for i, n in enumerate(names) if "affine_step_backward_kernel" in n)
lo = max(0, last - int(sys.argv[3]) - int(sys.argv[2]))
prev_end = None
for r in rows[lo:lo + int(sys.argv[3])]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%8.2f us  gap %7.2f  %s" % ((e - s) / 1e3, gap, r["Kernel_Name"][:110]))
    prev_end = e
PY
rm -rf $OUT/tl_$W
cat $OUT/timeline_$W.txt
