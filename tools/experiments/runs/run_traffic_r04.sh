#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# HBM / fabric traffic of the step's kernels (one TCC counter per pass, each under `timeout`): B/16 batch 32 and L/14 batch 16
# -> gpurun_out/r4_traffic.json (what bench.py quotes as roofline.traffic, bound to the kernel sources' digest), r4_hbm_traffic_{b16,l14}.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/r4_traffic.json
for wl in "owlvit-base-patch16 32 b16" "owlvit-large-patch14 16 l14"; do
  set -- $wl
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r4_pmc_${3}_$c -o p -f csv -- python $R/bench.py --arch $1 --batch $2 --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r4_pmc_${3}_$c.log 2>&1
  done
  (cd $R && python tools/pmc_traffic.py gpurun_out/r4_pmc_${3}_FETCH_SIZE gpurun_out/r4_pmc_${3}_WRITE_SIZE --json gpurun_out/r4_traffic.json --workload $1/$2 > gpurun_out/r4_hbm_traffic_$3.md)
  rm -rf $R/gpurun_out/r4_pmc_${3}_FETCH_SIZE $R/gpurun_out/r4_pmc_${3}_WRITE_SIZE
  head -14 $R/gpurun_out/r4_hbm_traffic_$3.md
done
cat $R/gpurun_out/r4_traffic.json
