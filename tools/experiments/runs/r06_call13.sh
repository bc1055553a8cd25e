#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 13: the 32 x Dt prompt-gradient product on the TN kernel -- tests, then same-box A/B through the Python switch
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "tn_slab or reference or flat_grad or trajectory or train_step or sub_batch" > gpurun_out/r6_c13_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_c13_tests.log
: > gpurun_out/r6_tn_small_n_ab.log
for round in 1 2 3; do
  for v in 0 1; do
    python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 7 --tn-small-n $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('round $round tn-small-n $v:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', c['window_values'])" >> gpurun_out/r6_tn_small_n_ab.log
  done
done
cat gpurun_out/r6_tn_small_n_ab.log
