#!/bin/bash
# Per pass of a workload's forward ELBO (eager warm-ups, then hipGraph replays): the mean gaps in front of the two launches
# of a timestep and their durations:
#   tools/timeline_graph.sh [workload] [rows]
set -u
W=${1:-c4b384}; ROWS=${2:-40}
OUT=gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tlg_$W -- \
   python $GRAFT_REPO_ROOT/bench.py --workload $W --mode graph --steps 2 --warmup 1 --no-cpu-baseline --extras off --no-backward > $GRAFT_REPO_ROOT/$OUT/tlg_$W.log 2>&1)
TRACE=$(ls $OUT/tlg_$W/*/*kernel_trace.csv | head -1)
python - "$TRACE" "$ROWS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
names = [r["Kernel_Name"] for r in rows]
step = lambda i: "affine_propagate" in names[i] or "affine_logweight_kernel" in names[i]
# every (resampling launch -> propagation launch -> next resampling launch) of the run, in passes of ~99: the mean gaps
# in front of the two launches and their durations (an eager pass shows the host's time per timestep in the gaps)
pairs = []
for i in range(1, len(rows)):
    if step(i) and "ancestor_index_inv" in names[i - 1]:
        s0, e0 = int(rows[i - 1]["Start_Timestamp"]), int(rows[i - 1]["End_Timestamp"])
        s1, e1 = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        before = int(rows[i - 2]["End_Timestamp"]) if i >= 2 else s0
        pairs.append(((s0 - before) / 1e3, (e0 - s0) / 1e3, (s1 - e0) / 1e3, (e1 - s1) / 1e3, s0))
passes, current = [], []
for p in pairs:
    if current and (p[4] - current[-1][4]) / 1e3 > 2000:      # a pause of 2 ms: the next pass
        passes.append(current)
        current = []
    current.append(p)
if current:
    passes.append(current)
mean = lambda xs: sum(xs) / max(1, len(xs))
for k, ps in enumerate(passes):
    print("pass %2d: %3d steps; gap before K2 %6.1f us, K2 %5.1f us, gap before propagation %6.1f us, propagation %6.1f us; "
          "per step %6.1f us" % (k, len(ps), mean([p[0] for p in ps]), mean([p[1] for p in ps]), mean([p[2] for p in ps]),
                                 mean([p[3] for p in ps]), mean([sum(p[:4]) for p in ps])))
PY
rm -rf $OUT/tlg_$W
