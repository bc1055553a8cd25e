#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5 hygiene: race screens of the kernels touched this round (rolling aux prefetch in gemm_pp2<8>, 32-bit / patch-14 gather in <7>, attn_cls_row) + the training soak
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1200 python tools/stress_pp.py 2>&1 | tail -12) > gpurun_out/r5_soak.log 2>&1
cat gpurun_out/r5_soak.log
