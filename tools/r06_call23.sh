#!/bin/bash
# round 6, GPU call 23: hungarian's column scan with its cost-row elements requested eight at a time -- old / new library, results + time; loss tests
cd "$(dirname "$0")/.."
[ -f ab_libs/libowlhip_old.so.bin ] && [ -f ab_libs/libowlhip_new.so.bin ] || { echo "needs ab_libs/libowlhip_{old,new}.so.bin"; exit 1; }
mkdir -p gpurun_out
L=gpurun_out/r6_hungarian_ab.log; : > $L
for round in 1 2; do for v in old new; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  echo "== $v (round $round)" >> $L
  python tools/hungarian_bench.py 2>&1 | grep -v amdgpu.ids >> $L
done; done
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
python -m pytest tests/test_loss_gpu.py -q -x 2>&1 | tail -3 >> $L
cat $L
