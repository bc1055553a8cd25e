#!/bin/bash
# round 6, GPU call 7: whole GPU suite + smoke on the final tree, tuning-build experiments' tests (then the product library is restored by the snapshot: nothing persists)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6_c7_tests.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r6_c7_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_c7_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r6_c7_smoke.log
