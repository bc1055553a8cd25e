#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 5 (tuning build in place): start stagger of the persistent GEMM
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
OWL_TUNING=1 timeout 900 python tools/experiments/stagger_ab.py > gpurun_out/r6_stagger_ab.log 2>&1; echo "rc=$?"; cut -c1-330 gpurun_out/r6_stagger_ab.log
