#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 9: new tests (phases mask), then the upper bound of VERDICT r05 #2(b): the train step with the dW chain's colsum / slab_reduce launches SKIPPED (timing only)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_determinism_gpu.py -q -m gpu -k "phases or pretransposed" > gpurun_out/r6_c9_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6_c9_tests.log
: > gpurun_out/r6_colsum_ablate.log
for round in 1 2; do
  for v in "" "colsum" "colsum,slab_reduce"; do
    python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 7 --ablate "$v" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('round $round skipped [$v]:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', c['window_values'])" >> gpurun_out/r6_colsum_ablate.log
  done
done
cat gpurun_out/r6_colsum_ablate.log
