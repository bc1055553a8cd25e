#!/bin/bash
# round 6, GPU call 5 (tuning build in place): start stagger of the persistent GEMM
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OWL_TUNING=1 timeout 900 python tools/experiments/stagger_ab.py > gpurun_out/r6_stagger_ab.log 2>&1; echo "rc=$?"; cut -c1-330 gpurun_out/r6_stagger_ab.log
