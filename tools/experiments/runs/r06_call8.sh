#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 8: the C-ABI all-reduce entry's test, then the whole profile set again on the final tree (the traffic file is bound to the source digest)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_rccl_gpu.py tests/test_abi.py -q -m gpu -k "allreduce or abi" > gpurun_out/r6_c8_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_c8_tests.log
bash tools/run_profiles_r06.sh > gpurun_out/r6_c8_profiles.log 2>&1; echo "profiles rc=$?"
head -22 gpurun_out/r6_c8_profiles.log | cut -c1-400
