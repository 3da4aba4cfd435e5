#!/bin/bash
# Runs on the GPU box (through gpurun): the two PMC passes over tools/pmc_gather.py, one counter per
# rocprofv3 run (kernel trace only, as the pool requires), then tools/pmc_summarize.py.
# Usage: tools/pmc_round.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv \
      -d $OUT/${TAG}_pmcdir_$C -- python $GRAFT_REPO_ROOT/tools/pmc_gather.py > $OUT/${TAG}_pmc_$C.log 2>&1)
  CSV=$(ls $OUT/${TAG}_pmcdir_$C/*/*counter_collection.csv | head -1)
  cp $CSV $OUT/${TAG}_pmc_${C}_raw.csv
  rm -rf $OUT/${TAG}_pmcdir_$C
  tail -2 $OUT/${TAG}_pmc_$C.log
done
python tools/pmc_summarize.py $OUT/${TAG}_pmc_FETCH_SIZE_raw.csv $OUT/${TAG}_pmc_WRITE_SIZE_raw.csv \
    $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv | tail -60
rm -f $OUT/${TAG}_pmc_*_raw.csv
