#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cp owl-vit-object-detection_amd/libowlhip.so /tmp/shipped.so
(python tools/attn_bench.py 2>&1 | tail -4) > gpurun_out/r5_attn_rs.log
cp ab_libs/libowlhip_tuning_rs.so.bin owl-vit-object-detection_amd/libowlhip.so
OWL_TUNING=1 python tools/attn_rowsum_mfma_ab.py >> gpurun_out/r5_attn_rs.log 2>&1
cp /tmp/shipped.so owl-vit-object-detection_amd/libowlhip.so
cat gpurun_out/r5_attn_rs.log
