#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5, GPU call 4: changed tests, then the small-batch lines on the caller-driven half-height rule (tile 6 through ops.gemm(concurrency=))
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_headline_gpu.py -m gpu -q -s -k "margin or small_problem or verdict_threshold or pingpong or patch_embed or headline or baseline_batch or training or step" > gpurun_out/r5_c4_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c4_tests.log
tail -6 gpurun_out/r5_c4_tests.log
: > gpurun_out/r5_c4_small.log
for round in 1 2; do
  for args in "--batch 1" "--batch 2" "--batch 4" "--batch 8" "--forward-only --batch 8" "--forward-only --batch 1" ""; do
    python bench.py --no-cpu-baseline --no-compare --steps 30 --warmup 5 $args 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('round $round [$args]', d['value'], 'img/s', d['ms_per_step'], 'ms;', [(r['kernel'][:14], r['ms_per_launch']) for r in [d['roofline']]+d['roofline_other']])" >> gpurun_out/r5_c4_small.log
  done
done
cat gpurun_out/r5_c4_small.log
python tools/batch_sweep.py 2>&1 | tail -8
