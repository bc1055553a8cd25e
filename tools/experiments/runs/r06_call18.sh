#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 18: counted waits behind EVERY epilogue + saved tile in the output's lane pairing + class head tail kernels pipelined -- kernel / determinism tests on the new
# library, then the headline bench (and L/14) with old / new library alternated
cd "$(dirname "$0")/../../.."
[ -f ab_libs/libowlhip_old.so.bin ] && [ -f ab_libs/libowlhip_new.so.bin ] || { echo "needs ab_libs/libowlhip_{old,new}.so.bin"; exit 1; }
mkdir -p gpurun_out
L=gpurun_out/r6_epilogue_waits_ab.log; : > $L
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py -q -x 2>&1 | tail -3 >> $L
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$1:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', c['window_values'], '; per launch', [(r['kernel'][:28], r['ms_per_launch']) for r in [d['roofline']] + d['roofline_other']])"; }
for round in 1 2 3; do for v in old new; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 7 2>/dev/null | tail -1 | line "round $round $v B/16" >> $L
done; done
for round in 1 2; do for v in old new; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
  python bench.py --arch owlvit-large-patch14 --batch 16 --steps 6 --warmup 2 --windows 3 --no-cpu-baseline --no-compare 2>/dev/null | tail -1 | line "round $round $v L/14" >> $L
done; done
cp ab_libs/libowlhip_new.so.bin owl-vit-object-detection_amd/libowlhip.so
cat $L
