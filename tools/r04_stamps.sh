#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
AESMC_PROBE_BUILD=1 AESMC_HIPCC_FLAGS="-DAESMC_K16_PROBES -DAESMC_LG_FAST_BUILD" python -m aesmc_amd.build --force > $OUT/r04_probe_build.txt 2>&1 || { tail -5 $OUT/r04_probe_build.txt; exit 1; }
timeout -k 10 300 python tools/k16stamps.py 1024 4096 10 "$@" > $OUT/r04_k16stamps.txt 2>&1; cat $OUT/r04_k16stamps.txt | tail -60
