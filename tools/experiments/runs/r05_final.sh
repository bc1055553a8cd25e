#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5: what the driver runs at round end -- the whole GPU suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r5_final_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_final_tests.log
tail -4 gpurun_out/r5_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>gpurun_out/r5_final_bench.err | tail -1 > gpurun_out/r5_final_bench.json; cut -c1-1200 gpurun_out/r5_final_bench.json
