#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5, GPU call 1: the whole GPU suite on the new build, then old (r04) / new library alternated under the dX-through-quick-GELU' timing and the bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r5_c1_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c1_tests.log
tail -5 gpurun_out/r5_c1_tests.log
AB_CMD="python tools/dqgelu_time.py" LIBS="old new" KEEP=new bash tools/ab_round4.sh > gpurun_out/r5_c1_ab.log 2>&1
cat gpurun_out/r5_c1_ab.log
