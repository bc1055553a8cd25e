#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 15: lag between the two sub-batch streams of the encoder (unlike kernels side by side) -- same-box A/B
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
: > gpurun_out/r6_stream_lag_ab.log
for round in 1 2; do
  for v in 0 2 3 4 6; do
    python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 7 --stream-lag $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('round $round stream-lag $v:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', c['window_values'])" >> gpurun_out/r6_stream_lag_ab.log
  done
done
cat gpurun_out/r6_stream_lag_ab.log
