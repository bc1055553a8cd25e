#!/bin/bash
# round 6, GPU call 16: class head tail kernels with loads in flight ahead of their use (class_sims fwd: 8..16 float4 per lane; bwd: rows 3 ahead) -- old / new library, bits + time
cd "$(dirname "$0")/.."
[ -f ab_libs/libowlhip_old.so.bin ] && [ -f ab_libs/libowlhip_new.so.bin ] || { echo "needs ab_libs/libowlhip_{old,new}.so.bin"; exit 1; }
mkdir -p gpurun_out
L=gpurun_out/r6_class_sims_ab.log; : > $L
for round in 1 2; do for v in old new; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  echo "== $v (round $round)" >> $L
  python tools/class_sims_bench.py >> $L 2>&1
done; done
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
python -m pytest tests/test_kernels_gpu.py -q -x -k "class_sims or box_final" 2>&1 | tail -5 >> $L
cat $L
