#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cp owl-vit-object-detection_amd/libowlhip.so /tmp/shipped.so
(python tools/attn_bench.py 2>&1 | tail -4) > gpurun_out/r5_attn_rs.log
cp ab_libs/libowlhip_tuning_rs.so.bin owl-vit-object-detection_amd/libowlhip.so
OWL_TUNING=1 python tools/attn_rowsum_mfma_ab.py >> gpurun_out/r5_attn_rs.log 2>&1
cp /tmp/shipped.so owl-vit-object-detection_amd/libowlhip.so
cat gpurun_out/r5_attn_rs.log
