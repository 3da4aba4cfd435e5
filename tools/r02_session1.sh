#!/bin/bash
# GPU session 1 of round 2 (through gpurun): parity of the changed kernels, the step for every
# parts value, the two sorted-backward kernels, and the default bench line.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -15 > $OUT/s1_pytest_kernels.txt
tail -4 $OUT/s1_pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_graphs.py -m gpu -x -q 2>&1 | tail -15 > $OUT/s1_pytest_infer.txt
tail -4 $OUT/s1_pytest_infer.txt
timeout 600 python tools/stepbench.py c2 c4s c4 > $OUT/s1_stepbench.txt 2>&1
cat $OUT/s1_stepbench.txt
timeout 600 python tools/kbench.py c2 c4 > $OUT/s1_kbench.txt 2>&1
grep -i "backward\|==" $OUT/s1_kbench.txt
timeout 1200 python bench.py > $OUT/s1_bench_default.json 2> $OUT/s1_bench_default.err
tail -5 $OUT/s1_bench_default.err
python -c "
import json
d=json.load(open('$OUT/s1_bench_default.json'))
print(json.dumps({k:v for k,v in d.items() if k not in ('kernels','extras')}, indent=1))
for k,v in d.get('extras',{}).items(): print(k, json.dumps(v)[:1500])
"
