"""Round 4, tuning build, timing only: what the shipped two-phase GEMM (csrc/gemm_pp2.hip) spends on its LDS-DMA requests, its fragment reads and its epilogue --
the same ablations as tools/gemm_fr_ablate.py (results are wrong; only the clock counts).   OWL_TUNING=1 python tools/gemm_pp2_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"


def t(fn, iters=10, rounds=3):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(out)[len(out) // 2]


M = 32 * 2312
for name, N, K, epi in (("warm", 2304, 768, ops.EPI_BIAS_BF16), ("QKV", 2304, 768, ops.EPI_BIAS_BF16), ("fc1", 3072, 768, ops.EPI_QGELU_BF16), ("fc2", 768, 3072, ops.EPI_BIAS_BF16)):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); b = torch.randn(N, device=DEV)
    o = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
    res = {}
    for abl, label in ((0, "shipped"), (4, "no epilogue"), (1, "no DMA"), (2, "no fragment reads"), (5, "no DMA, no epilogue"), (7, "MFMAs + barriers only")):
        _lib.call("owl_gemm_pp2_ablate", abl)
        res[label] = t(lambda: ops.gemm(epi, A, W, o, bias=b, M=M, tile=7))      # tile 7: the two-phase kernel on the whole problem (no remainder launch)
    _lib.call("owl_gemm_pp2_ablate", 0)
    if name != "warm":
        print(f"{name:4s} N={N} K={K}: " + " | ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
