#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 10: tall reduction of box_final_bwd's partials -- tests, then same-box A/B against the previous library (two processes alternated)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py -x -q -m gpu -k "slab or box or reference or flat_grad or train_step" > gpurun_out/r6_c10_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6_c10_tests.log
: > gpurun_out/r6_tall_reduce_ab.log
for round in 1 2 3; do
  for v in old new; do
    cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
    python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 7 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('round $round library $v:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', c['window_values'])" >> gpurun_out/r6_tall_reduce_ab.log
  done
done
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
cat gpurun_out/r6_tall_reduce_ab.log
