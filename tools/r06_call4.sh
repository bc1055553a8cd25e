#!/bin/bash
# round 6, GPU call 4 (tuning build in place): dX through quick-GELU' -- quad-contiguous aux loads + stores (25 spilled VGPRs) against the accumulator-layout form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OWL_TUNING=1 timeout 600 python tools/experiments/dqgelu_lines_ab.py > gpurun_out/r6_dqgelu_lines_ab.log 2>&1; echo "rc=$?"; cut -c1-300 gpurun_out/r6_dqgelu_lines_ab.log
OWL_TUNING=1 timeout 600 python tools/experiments/dqgelu_lines_ab.py cold >> gpurun_out/r6_dqgelu_lines_ab.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r6_dqgelu_lines_ab.log | cut -c1-300
