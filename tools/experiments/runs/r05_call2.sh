#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5, GPU call 2: the new / changed tests, then the GEMM counter table
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_tokenizer.py tests/test_ddp_rccl_gpu.py tests/test_determinism_gpu.py -m gpu -q -s -k "margin or trained_like or vocab or bench or determin or bitwise" > gpurun_out/r5_c2_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c2_tests.log
tail -8 gpurun_out/r5_c2_tests.log
bash tools/r05_gemm_counters.sh
