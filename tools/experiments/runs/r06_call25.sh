#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 25: race screens of the shipped build (GEMM incl. the quick-GELU epilogue with its saved tile; attention) under concurrent HBM traffic + the training soak
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
(timeout 1200 python tools/stress_pp.py 2>&1 | grep -v amdgpu.ids | tail -14) > gpurun_out/r6_stress.log 2>&1
(timeout 900 python tools/stress_attn.py 2>&1 | grep -v amdgpu.ids | tail -8) >> gpurun_out/r6_stress.log 2>&1
(timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -8) >> gpurun_out/r6_stress.log 2>&1
cat gpurun_out/r6_stress.log
