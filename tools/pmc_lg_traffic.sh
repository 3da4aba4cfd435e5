#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per run) of the linear-Gaussian kernels at
# B=1024 K=4096 d=10 over tools/pmc_lg.py; writes gpurun_out/pmc_lg_traffic.csv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmclgt_$C -- \
     python $GRAFT_REPO_ROOT/tools/pmc_lg.py > $OUT/pmclgt_$C.log 2>&1)
  CSV=$(ls $OUT/pmclgt_$C/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/pmclgt_$C.csv
  rm -rf $OUT/pmclgt_$C
done
python - <<PY
import csv, collections
out = collections.OrderedDict()
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("$OUT/pmclgt_%s.csv" % counter)):
        name = r.get("Kernel_Name", "")
        if "aesmc::" not in name or r.get("Counter_Name") != counter:
            continue
        key = name.split("(")[0].replace("void ", "")
        out.setdefault(key, {}).setdefault(counter, []).append(float(r["Counter_Value"]) * 1024.0)
with open("$OUT/pmc_lg_traffic.csv", "w") as fh:
    fh.write("kernel,launches,fetch_MB_raw,fetch_MB_x1.99,write_MB,hbm_MB\n")
    for key, v in out.items():
        f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0])))
        w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
        fh.write('"%s",%d,%.1f,%.1f,%.1f,%.1f\n' % (key, len(v.get("FETCH_SIZE", [])), f / 1e6, 1.99 * f / 1e6, w / 1e6, (1.99 * f + w) / 1e6))
print(open("$OUT/pmc_lg_traffic.csv").read())
PY
rm -f $OUT/pmclgt_*.log
