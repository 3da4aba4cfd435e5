#!/bin/bash
# A whole-ELBO hipGraph at B=512 K=4096 (one GPU's shard at N=2), forward only, with the runtime's packet-capture fast
# path on and off: replay time and the loss against the eager loop's.
set -u
OUT=gpurun_out
for PC in 1 0; do
  for i in 1 2; do
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=$PC timeout -k 10 300 python bench.py --workload c4x2 --mode graph --steps 10 --warmup 3 --extras off --no-cpu-baseline --no-backward > $OUT/r04_pc${PC}_$i.json 2> $OUT/r04_pc${PC}_$i.err
    python -c "
import json; d=json.loads(open('$OUT/r04_pc${PC}_$i.json').read().strip().splitlines()[-1]); print('packet capture $PC run $i:', d['mode'], round(d['ms_per_step'],3), 'loss', d['loss'], d.get('graph_error'))" || tail -3 $OUT/r04_pc${PC}_$i.err
  done
done
timeout -k 10 300 python bench.py --workload c4x2 --mode eager --steps 10 --warmup 3 --extras off --no-cpu-baseline --no-backward > $OUT/r04_eager.json 2>/dev/null
python -c "
import json; d=json.loads(open('$OUT/r04_eager.json').read().strip().splitlines()[-1]); print('eager:', round(d['ms_per_step'],3), 'loss', d['loss'])"
