#!/bin/bash
# builds the library with the K14 probe switches compiled in (on the GPU box, into its own copy) and times the launch
set -u
OUT=gpurun_out; mkdir -p $OUT
AESMC_PROBE_BUILD=1 AESMC_HIPCC_FLAGS="-DAESMC_K14_PROBES -DAESMC_LG_FAST_BUILD" python -m aesmc_amd.build --force > $OUT/r04_k14probe_build.txt 2>&1 || { tail -5 $OUT/r04_k14probe_build.txt; exit 1; }
timeout -k 10 400 python tools/k14probe.py "$@" > $OUT/r04_k14probe.txt 2>&1; cat $OUT/r04_k14probe.txt | tail -20
