#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# The round's profile set, run on the GPU box through gpurun (outputs under gpurun_out/, summaries copied to profiles/ by hand).
#   1. rocprofv3 --kernel-trace --stats over the default bench      -> prof_final/  (+ tools/prof_summary.py table)
#   2. the un-profiled default bench (with cpu_baseline)             -> bench_final.json
#   3. PMC passes over tools/pmc_kernels.py, ONE pass per counter group and ONE TCC counter per pass (FETCH_SIZE and
#      WRITE_SIZE together exceed the hardware's counter capacity and make rocprofv3 abort and hang); never combined with
#      trace domains other than --kernel-trace; every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$1" != "pmc-only" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o rf -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_final_bench.log 2>&1
  timeout 600 python $R/bench.py > $R/gpurun_out/bench_final.log 2>&1; tail -1 $R/gpurun_out/bench_final.log > $R/gpurun_out/bench_final.json
fi
[ "$1" = "no-pmc" ] && { cd $R; python tools/prof_summary.py $(ls gpurun_out/prof_final/*.db | head -1) 45 > gpurun_out/prof_final_summary.md; exit 0; }
i=0
for C in "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  [ -n "$2" ] && [ "$i" -lt "$2" ] && continue
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_final_$i -o p -f csv -- python $R/tools/pmc_kernels.py > $R/gpurun_out/pmc_final_$i.log 2>&1
done
cd $R
[ "$1" != "pmc-only" ] && python tools/prof_summary.py $(ls gpurun_out/prof_final/*.db | head -1) 45 > gpurun_out/prof_final_summary.md
true
