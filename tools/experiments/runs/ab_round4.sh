#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# Same-box A/B of two builds of libowlhip.so (ab_libs/libowlhip_old.so.bin / libowlhip_new.so.bin, made in the build container), alternated twice:
# $AB_CMD (default: the narrow-output GEMM timing) and the bench.  LIBS="old new ..." names the builds, KEEP (default new) is left in place.
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2; do for v in ${LIBS:-old new}; do
  cp ab_libs/libowlhip_$v.so.bin owl-vit-object-detection_amd/libowlhip.so; echo "== $v (round $round)"
  ${AB_CMD:-python tools/gemm_auto_time.py} 2>&1 | grep -v amdgpu.ids | tail -6
  python bench.py --no-cpu-baseline --no-compare --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'img/s;', [(r['kernel'][:14], r['ms_per_launch']) for r in [d['roofline']]+d['roofline_other']])"
done; done
cp ab_libs/libowlhip_${KEEP:-new}.so.bin owl-vit-object-detection_amd/libowlhip.so
