#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 21: box_final with a wave walking many rows (weights in registers, one butterfly for the four sums) + the image cast four pieces per trip -- old / new library, bits + time
cd "$(dirname "$0")/../../.."
[ -f ab_libs/libowlhip_old.so.bin ] && [ -f ab_libs/libowlhip_new.so.bin ] || { echo "needs ab_libs/libowlhip_{old,new}.so.bin"; exit 1; }
mkdir -p gpurun_out
L=gpurun_out/r6_heads_fwd_ab.log; : > $L
for round in 1 2; do for v in old new; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  echo "== $v (round $round)" >> $L
  python tools/heads_fwd_bench.py 2>&1 | grep -v amdgpu.ids >> $L
done; done
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
python -m pytest tests/test_kernels_gpu.py -q -x -k "box_final or cast" 2>&1 | tail -3 >> $L
cat $L
