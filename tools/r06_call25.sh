#!/bin/bash
# round 6, GPU call 25: race screens of the shipped build (GEMM incl. the quick-GELU epilogue with its saved tile; attention) under concurrent HBM traffic + the training soak
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 1200 python tools/stress_pp.py 2>&1 | grep -v amdgpu.ids | tail -14) > gpurun_out/r6_stress.log 2>&1
(timeout 900 python tools/stress_attn.py 2>&1 | grep -v amdgpu.ids | tail -8) >> gpurun_out/r6_stress.log 2>&1
(timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -8) >> gpurun_out/r6_stress.log 2>&1
cat gpurun_out/r6_stress.log
