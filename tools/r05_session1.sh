#!/bin/bash
# Round 5, first GPU session: the full -m gpu suite on the round's first tree, the graph fast-path question at the
# 8-GPU shard (DEBUG_CLR_GRAPH_PACKET_CAPTURE 0 / 1), the step's launches at the shard sizes, configs[4] as BASELINE
# states it (c5) and its healthy variant (c5h) with the CPU baseline beside them.
set -u
OUT=gpurun_out; mkdir -p $OUT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "step killed at its limit: stopping"; exit $1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu --maxfail=8 -q > $OUT/r05a_pytest_gpu.txt 2>&1; rc=$?
tail -5 $OUT/r05a_pytest_gpu.txt | cut -c1-300
[ $rc -ne 0 ] && grep -n "Error\|assert\|FAILED" $OUT/r05a_pytest_gpu.txt | head -40 | cut -c1-300
stop_if_killed $rc
for PC in 0 1; do
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=$PC timeout -k 10 300 python bench.py --workload c4s --steps 10 --warmup 3 --no-cpu-baseline \
      --extras off --no-backward > $OUT/r05a_c4s_pc$PC.json 2> $OUT/r05a_c4s_pc$PC.err; rc=$?
  stop_if_killed $rc
  python - <<PY
import json
try:
    d = json.load(open("$OUT/r05a_c4s_pc$PC.json"))
    print("c4s packet capture $PC:", round(d["ms_per_step"], 3), "ms", d["mode"], d.get("graph_error"))
except Exception as e:
    print("c4s pc$PC: no line", e)
PY
done
for B in 128 256; do
  timeout -k 10 300 python tools/k16bench.py $B 4096 10 > $OUT/r05a_k16bench_B$B.txt 2>&1; rc=$?; stop_if_killed $rc
  cat $OUT/r05a_k16bench_B$B.txt | cut -c1-150
done
for W in c5h c5; do
  timeout -k 10 420 python bench.py --workload $W --steps 5 --warmup 2 --extras off > $OUT/r05a_bench_$W.json 2> $OUT/r05a_bench_$W.err; rc=$?
  stop_if_killed $rc
  python - <<PY
import json
try:
    d = json.load(open("$OUT/r05a_bench_$W.json"))
    print("$W:", round(d["ms_per_step"], 2), "ms", "%.3g" % d["value"], d["roofline"]["kernel"][:40], d["roofline"]["bound"], d["roofline"]["frac"],
          "cpu", "%.3g" % d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["seconds"])
except Exception as e:
    print("$W: no line", e)
PY
done
