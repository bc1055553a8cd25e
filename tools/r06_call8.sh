#!/bin/bash
# round 6, GPU call 8: the C-ABI all-reduce entry's test, then the whole profile set again on the final tree (the traffic file is bound to the source digest)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_rccl_gpu.py tests/test_abi.py -q -m gpu -k "allreduce or abi" > gpurun_out/r6_c8_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_c8_tests.log
bash tools/run_profiles_r06.sh > gpurun_out/r6_c8_profiles.log 2>&1; echo "profiles rc=$?"
head -22 gpurun_out/r6_c8_profiles.log | cut -c1-400
