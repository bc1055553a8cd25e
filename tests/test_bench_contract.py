"""CPU-side checks of bench.py's bookkeeping (no GPU): the counter-traffic file it quotes as `roofline.traffic` was measured on THESE kernel sources, and the
`cpu_baseline` leg produces the fields the contract names."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench_mod():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        sys.path.insert(0, ROOT)
        import bench
        return bench
    finally:
        sys.argv = argv


def test_traffic_file_is_bound_to_the_shipped_kernel_sources(bench_mod):
    """`roofline.traffic` is a PMC figure taken in separate rocprofv3 passes (tools/run_profiles_r06.sh) and read from profiles/: bench.py refuses it (traffic =
    null) when the kernel sources changed since.  This test makes that refusal visible BEFORE a round ends: edit a kernel -> re-run the traffic passes."""
    t = json.load(open(os.path.join(ROOT, bench_mod.TRAFFIC_FILE)))
    assert t["kernel_source_digest"] == bench_mod.kernel_source_digest(), (
        f"{bench_mod.TRAFFIC_FILE} was measured on other kernel sources: re-run tools/run_profiles_r06.sh on a GPU box and copy gpurun_out/r6_traffic.json over it")
    assert bench_mod.TRAFFIC_NOTE is None
    for wl in ("owlvit-base-patch16/32", "owlvit-large-patch14/16"):
        assert set(t["workloads"][wl]) == {bench_mod.LABEL_BIAS, bench_mod.LABEL_QGELU, bench_mod.LABEL_ATTN}
        assert all(1e8 < v < 5e9 for v in t["workloads"][wl].values())


def test_cpu_baseline_leg_fields(bench_mod):
    from owl_vit_object_detection_amd.config import get_config
    out = bench_mod.cpu_baseline(get_config("tiny"), 5)
    assert out["kind"] == "port" and out["unit"] == "images/sec" and out["value"] > 0 and 1 <= out["cores"] <= (os.cpu_count() or 1)
    assert "AdamW" in out["sample"] and "median" in out["sample"]
