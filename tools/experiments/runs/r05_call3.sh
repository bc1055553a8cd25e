#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5, GPU call 3: whole GPU suite on the new build (patch-14 gather, ABI 5, small-problem rule), then the small-problem rule A/B at batch 1 / 2 / 8
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r5_c3_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c3_tests.log
tail -12 gpurun_out/r5_c3_tests.log
: > gpurun_out/r5_c3_small_ab.log
for round in 1 2; do for v in nosmall small; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  for args in "--batch 1" "--batch 2" "--batch 4 --encoder-streams 1" "--forward-only --batch 8" "--forward-only --batch 1"; do
    python bench.py --no-cpu-baseline --no-compare --steps 30 --warmup 5 $args 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v round $round [$args]', d['value'], 'img/s', d['ms_per_step'], 'ms;', [(r['kernel'][:14], r['ms_per_launch']) for r in [d['roofline']]+d['roofline_other']])" >> gpurun_out/r5_c3_small_ab.log
  done
done; done
cp ab_libs/libowlhip_small.so.bin owl-vit-object-detection_amd/libowlhip.so
cat gpurun_out/r5_c3_small_ab.log
