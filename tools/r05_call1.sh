#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the new build, then old (r04) / new library alternated under the dX-through-quick-GELU' timing and the bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r5_c1_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c1_tests.log
tail -5 gpurun_out/r5_c1_tests.log
AB_CMD="python tools/dqgelu_time.py" LIBS="old new" KEEP=new bash tools/ab_round4.sh > gpurun_out/r5_c1_ab.log 2>&1
cat gpurun_out/r5_c1_ab.log
