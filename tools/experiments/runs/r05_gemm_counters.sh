#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 5: the table that predicts a GEMM's fabric traffic from its tile walk -- per GEMM of the step and column-block width: time (no profiler),
# then one TCC counter per rocprofv3 pass (FETCH_SIZE, TCC_HIT_sum, TCC_MISS_sum, TCC_EA0_RDREQ_sum).  Tuning build (ab_libs/libowlhip_tuning.so.bin).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export OWL_TUNING=1
cp $R/owl-vit-object-detection_amd/libowlhip.so /tmp/libowlhip_shipped.so
cp $R/ab_libs/libowlhip_tuning.so.bin $R/owl-vit-object-detection_amd/libowlhip.so
OUT=$R/gpurun_out/r5_gemm_counters.log; : > $OUT
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' >> $OUT; echo >> $OUT
for cfg in "fc1 2" "fc1 3" "fc1 4" "fc1 6" "fc1 12" "qkv 3" "qkv 9" "dqgelu 0" "fc2 3"; do
  set -- $cfg
  (cd $R && timeout 120 python tools/gemm_counters.py $1 $2 7 2>&1 | grep "us per launch") >> $OUT
  for c in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
    d=/tmp/pmc_$1_$2_$c; rm -rf $d
    (cd $R && timeout 180 rocprofv3 --kernel-trace --pmc $c -d $d -o p -f csv -- python tools/gemm_counters.py $1 $2 4 > /dev/null 2>&1)
    python - "$d" "$c" "$1 bw=$2" >> $OUT <<'PY'
import csv, glob, os, sys
d, c, tag = sys.argv[1:4]
vals = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == c and "gemm_pp" in row["Kernel_Name"]:
            vals.append(float(row["Counter_Value"]))
vals = vals[2:]          # (the two warm-up launches)
print(f"   {tag} {c}: " + (f"{sum(vals) / len(vals):.0f} per launch ({len(vals)} launches)" if vals else "no data"))
PY
    rm -rf $d
  done
done
cp /tmp/libowlhip_shipped.so $R/owl-vit-object-detection_amd/libowlhip.so
cat $OUT
