#!/bin/bash
# Counter passes over the linear-Gaussian kernels; writes gpurun_out/pmc_lg_counters.csv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmclg_$i -- \
     python $GRAFT_REPO_ROOT/tools/pmc_lg.py > $OUT/pmclg_$i.log 2>&1)
  CSV=$(ls $OUT/pmclg_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/pmclg_set$i.csv
  rm -rf $OUT/pmclg_$i
done
python tools/pmc_kernel_table.py $OUT/pmc_lg_counters.csv aesmc:: $(ls $OUT/pmclg_set*.csv)
rm -f $OUT/pmclg_set*.csv
