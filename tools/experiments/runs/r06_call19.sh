#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 19: per-kernel durations (rocprofv3 kernel trace, one-stream schedule) of the headline step under the old and the new library
cd "$(dirname "$0")/../../.."
[ -f ab_libs/libowlhip_old.so.bin ] && [ -f ab_libs/libowlhip_new.so.bin ] || { echo "needs ab_libs/libowlhip_{old,new}.so.bin"; exit 1; }
R=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for v in old new; do
  cp $R/ab_libs/libowlhip_$v.so.bin $R/owl-vit-object-detection_amd/libowlhip.so
  rm -rf $R/gpurun_out/r6_prof_$v
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_prof_$v -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --windows 1 > $R/gpurun_out/r6_prof_$v.log 2>&1
  (cd $R && python tools/prof_summary.py $(ls gpurun_out/r6_prof_$v/*.db | head -1) 70 > gpurun_out/r6_prof_${v}_summary.md; rm -rf gpurun_out/r6_prof_$v)
done
cp $R/ab_libs/libowlhip_new.so.bin $R/owl-vit-object-detection_amd/libowlhip.so
cd $R; head -75 gpurun_out/r6_prof_new_summary.md | cut -c1-160
