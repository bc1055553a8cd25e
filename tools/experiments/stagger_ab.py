"""Round 6, tuning build: start stagger of the persistent ping-pong GEMM (owl_gemm_pp2_stagger(n): every second workgroup of an XCD starts n x ~8 k cycles late).
Hypothesis: all 256 CUs reach their epilogues together and burst 32 MB of stores (+ 32 MB of loads for dX through quick-GELU') at the HBM at once (5.5 us per 128-KiB
tile with 256 CUs storing, 1.5 us with 64: tools/probe/store_pattern.hip); two half-populations half a tile apart would spread the bursts.  One process,
alternating, 10 back-to-back launches per sample (a launch's own stagger tail included), bitwise check against stagger 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from owl_vit_object_detection_amd import _lib, ops
assert _lib.is_tuning_build(), "needs the tuning build (tools/experiments/build.sh)"
M = 32 * 2312
SHAPES = [("QKV", M, 2304, 768, ops.EPI_BIAS_BF16), ("out-proj", M, 768, 768, ops.EPI_BIAS_BF16), ("fc1 quick-GELU", M, 3072, 768, ops.EPI_QGELU_BF16),
          ("fc2", M, 768, 3072, ops.EPI_BIAS_BF16), ("dX through quick-GELU'", M, 3072, 768, ops.EPI_DQGELU_BF16), ("half-batch QKV", M // 2, 2304, 768, ops.EPI_BIAS_BF16),
          ("L/14 fc1", 16 * 3608, 4096, 1024, ops.EPI_QGELU_BF16)]
STAG = [0, 1, 2, 3, 4]
for name, m, N, K, epi in SHAPES:
    torch.manual_seed(1)
    A = torch.randn(ops.pad_rows(m), K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = None if epi == ops.EPI_DQGELU_BF16 else torch.randn(N, device="cuda")
    aux = torch.randn(ops.pad_rows(m), N, device="cuda").bfloat16() if epi == ops.EPI_DQGELU_BF16 else None
    o = torch.zeros(ops.pad_rows(m), N, device="cuda", dtype=torch.bfloat16)
    run = lambda: ops.gemm(epi, A, W, o, bias=bias, aux=aux, M=m)
    _lib.call("owl_gemm_pp2_stagger", 0); run(); torch.cuda.synchronize(); ref = o.clone()
    res = {s: [] for s in STAG}; same = {}
    for rnd in range(3):
        for s in STAG:
            _lib.call("owl_gemm_pp2_stagger", s)
            o.zero_(); run(); torch.cuda.synchronize(); same[s] = torch.equal(o, ref)
            ts = []
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            res[s].append(float(np.median(ts)))
    _lib.call("owl_gemm_pp2_stagger", 0)
    med = {s: float(np.median(v)) for s, v in res.items()}
    print(f"{name:24s} " + "  ".join(f"stagger {s}: {med[s]:7.1f} us ({100 * (med[s] / med[0] - 1):+.1f} %)" for s in STAG) + f"   bitwise: {all(same.values())}", flush=True)
