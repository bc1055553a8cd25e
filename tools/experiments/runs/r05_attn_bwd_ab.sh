#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# timing-only upper bound (round 5): attention backward without its four "constant" MFMAs per tile (-lse / -D broadcast through the matrix pipe)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; : > gpurun_out/r5_attn_bwd_ab.log
for round in 1 2; do for v in shipped noconst; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  echo "== $v (round $round)" >> gpurun_out/r5_attn_bwd_ab.log
  python tools/attn_bwd_bench.py 2>&1 | grep "attention bwd" | tail -2 >> gpurun_out/r5_attn_bwd_ab.log
done; done
cp ab_libs/libowlhip_shipped.so.bin owl-vit-object-detection_amd/libowlhip.so
cat gpurun_out/r5_attn_bwd_ab.log
