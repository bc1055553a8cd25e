#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 2: input stage after the staging rewrite (tests + bench), quick-GELU rate probe, one-stream kernel trace with a launch-by-launch listing
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_prefetcher.py -x -q -m gpu > gpurun_out/r6_c2_prefetch_tests.log 2>&1; echo "prefetch tests rc=$?"; tail -3 gpurun_out/r6_c2_prefetch_tests.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r6_c2_bench.json 2> gpurun_out/r6_c2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6_c2_bench.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], "windows", c.get("window_values"), "spread", c.get("window_spread_pct"))
    for k in c:
        if "host" in k or "h2d" in k or k == "gpu_seconds":
            print(" ", k, c[k] if not isinstance(c[k], str) else c[k][:60])
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r6_c2_bench.err").read()[-3000:])
PY
timeout 300 tools/probe/qgelu_rate > gpurun_out/r6_qgelu_rate.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r6_qgelu_rate.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_prof_b16 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --windows 1 > $R/gpurun_out/r6_prof_b16.log 2>&1; echo "rocprof rc=$?"
cd $R
DB=$(ls gpurun_out/r6_prof_b16/*.db 2>/dev/null | head -1)
python tools/step_listing.py $DB 5 --list > gpurun_out/r6_step_listing_b16.txt 2>&1
python tools/prof_summary.py $DB 60 > gpurun_out/r6_prof_b16_summary.md 2>&1
rm -rf gpurun_out/r6_prof_b16/*.db
head -70 gpurun_out/r6_step_listing_b16.txt
