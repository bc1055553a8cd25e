#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 7: whole GPU suite + smoke on the final tree, tuning-build experiments' tests (then the product library is restored by the snapshot: nothing persists)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6_c7_tests.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r6_c7_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_c7_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r6_c7_smoke.log
