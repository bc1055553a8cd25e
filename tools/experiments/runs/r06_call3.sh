#!/bin/bash
# ARCHIVED (end of round 6): the record of a gpurun call of this round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that moment; some copy untracked
# library builds (ab_libs/*.so.bin) over the shipped libowlhip.so, some use bench.py flags that were removed after the measurement.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# round 6, GPU call 3: stored quick-GELU derivative (kernel tests, model tests, determinism), attention-backward overlap A/B, bench B/16 + L/14
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py tests/test_headline_gpu.py tests/test_training_gpu.py -x -q -m gpu > gpurun_out/r6_c3_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6_c3_tests.log
timeout 600 python tools/attn_bwd_overlap_ab.py > gpurun_out/r6_attn_bwd_overlap.log 2>&1; echo "overlap rc=$?"; cat gpurun_out/r6_attn_bwd_overlap.log | cut -c1-260
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-compare > gpurun_out/r6_c3_bench.json 2> gpurun_out/r6_c3_bench.err; echo "bench rc=$?"
timeout 900 python bench.py --arch owlvit-large-patch14 --batch 16 --steps 6 --warmup 2 --windows 3 --no-cpu-baseline --no-compare > gpurun_out/r6_c3_bench_l14.json 2> gpurun_out/r6_c3_bench_l14.err; echo "bench l14 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r6_c3_bench.json", "gpurun_out/r6_c3_bench_l14.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "windows", d["config"].get("window_values"))
        for r in [d["roofline"]] + d["roofline_other"]:
            print("   ", r["kernel"][:60], r["ms_per_launch"], r["frac"])
    except Exception as e:
        print(f, "parse failed", e)
PY
