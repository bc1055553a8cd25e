#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5: train-step and forward-only throughput by batch size on one box (B/16 768^2), + the default line once more (box spread)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; : > gpurun_out/r5_batch_curve.log
for b in 1 2 4 8 16 32 64; do
  for mode in "" "--forward-only"; do
    python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 4 --batch $b $mode 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch $b [$mode]', d['value'], 'img/s', d['ms_per_step'], 'ms/step, step_mfma_frac', d['config']['step_mfma_frac'])" >> gpurun_out/r5_batch_curve.log
  done
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900 >> gpurun_out/r5_batch_curve.log
cat gpurun_out/r5_batch_curve.log
