#!/bin/bash
# round 5 hygiene: race screens of the kernels touched this round (rolling aux prefetch in gemm_pp2<8>, 32-bit / patch-14 gather in <7>, attn_cls_row) + the training soak
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1200 python tools/stress_pp.py 2>&1 | tail -12) > gpurun_out/r5_soak.log 2>&1
cat gpurun_out/r5_soak.log
