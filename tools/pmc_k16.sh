#!/bin/bash
# Counter passes over the forward step's launches; writes gpurun_out/pmc_k16_counters.csv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmck16_$i -- \
     python $GRAFT_REPO_ROOT/tools/pmc_k16.py > $OUT/pmck16_$i.log 2>&1)
  CSV=$(ls $OUT/pmck16_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $OUT/pmck16_set$i.csv
  tail -2 $OUT/pmck16_$i.log
  rm -rf $OUT/pmck16_$i
done
python tools/pmc_kernel_table.py $OUT/pmc_k16_counters.csv aesmc:: $(ls $OUT/pmck16_set*.csv)
rm -f $OUT/pmck16_set*.csv
cat $OUT/pmc_k16_counters.csv
