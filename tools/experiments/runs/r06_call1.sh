#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 1: new input-stage tests, whole GPU suite, smoke, headline bench (windows + product input stage)
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prefetcher.py tests/test_preprocess.py -x -q -m gpu > gpurun_out/r6_c1_prefetch_tests.log 2>&1; echo "prefetch tests rc=$?"
tail -3 gpurun_out/r6_c1_prefetch_tests.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6_c1_tests.log 2>&1; echo "suite rc=$?"
tail -3 gpurun_out/r6_c1_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_c1_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6_c1_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r6_c1_bench.json 2> gpurun_out/r6_c1_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6_c1_bench.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("value", d["value"], "ms", d["ms_per_step"], "windows", c.get("window_values"), "spread", c.get("window_spread_pct"))
    for k in c:
        if "host" in k or "h2d" in k or k == "gpu_seconds":
            print(" ", k, c[k] if not isinstance(c[k], str) else c[k][:60])
    print("roofline", d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r6_c1_bench.err").read()[-3000:])
PY
