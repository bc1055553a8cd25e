#!/bin/bash
# round 6, GPU call 15: lag between the two sub-batch streams of the encoder (unlike kernels side by side) -- same-box A/B
cd "$(dirname "$0")/.."
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
