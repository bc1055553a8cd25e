#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# HBM / fabric traffic of the step's kernels only (the two TCC passes of tools/run_profiles_r03.sh) -> gpurun_out/r3_traffic.json, r3_hbm_traffic.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r3_pmc_fetch -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r3_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r3_pmc_write -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r3_pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/r3_pmc_fetch gpurun_out/r3_pmc_write --json gpurun_out/r3_traffic.json > gpurun_out/r3_hbm_traffic.md
rm -f gpurun_out/r3_pmc_fetch/*kernel_trace.csv gpurun_out/r3_pmc_write/*kernel_trace.csv; gzip -f gpurun_out/r3_pmc_fetch/*counter_collection.csv gpurun_out/r3_pmc_write/*counter_collection.csv 2>/dev/null
head -12 gpurun_out/r3_hbm_traffic.md; cat gpurun_out/r3_traffic.json
