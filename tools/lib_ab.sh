#!/bin/bash
# Same-box A/B of two builds of libowlhip.so (ab_libs/libowlhip_old.so.bin / libowlhip_new.so.bin, made in the build container): alternates them under
# the stand-alone GEMM timing and the bench (separate processes, two rounds).  LIBS="old newA newB" names the builds; leaves $KEEP (default: new) in place.
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2; do
  for v in ${LIBS:-old new}; do
    cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so
    echo "== $v (round $round)"
    python tools/gemm_launch_scaling.py 2>&1 | grep "^M=" | sed 's/us per call by number of back-to-back calls -> //; s/ 1: .* 50: / 50 calls: /' | tr '\n' ';'; echo
    python bench.py --no-cpu-baseline --no-compare --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms;', [(r['kernel'], r['ms_per_launch']) for r in [d['roofline']]+d['roofline_other']])"
  done
done
cp ab_libs/libowlhip_${KEEP:-new}.so.bin owl-vit-object-detection_amd/libowlhip.so
