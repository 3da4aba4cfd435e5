#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
bash tools/prof_one.sh c2 > $OUT/r03g_prof_c2.txt 2>&1; cat $OUT/r03g_prof_c2.txt
cp $OUT/one_rocprof_c2.csv $OUT/r03g_rocprof_c2.csv
bash tools/prof_one.sh c4 > $OUT/r03g_prof_c4.txt 2>&1; cat $OUT/r03g_prof_c4.txt
cp $OUT/one_rocprof_c4.csv $OUT/r03g_rocprof_c4.csv
