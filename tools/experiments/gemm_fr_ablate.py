"""Round 4, tuning build, timing only: what bounds the K-loop of the free-running GEMM (csrc/gemm_fr.hip)?  The kernel without its LDS-DMA requests, without its
fragment reads, without both (results are wrong; only the clock counts), against the full kernel and the shipped one.   OWL_TUNING=1 python tools/gemm_fr_ablate.py"""
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
for name, N, K, epi in (("QKV", 2304, 768, ops.EPI_BIAS_BF16), ("fc1", 3072, 768, ops.EPI_QGELU_BF16), ("fc2", 768, 3072, ops.EPI_BIAS_BF16)):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); b = torch.randn(N, device=DEV)
    o = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
    res = {"shipped": t(lambda: ops.gemm(epi, A, W, o, bias=b, M=M, tile=0))}
    ref = torch.zeros_like(o); ops.gemm(epi, A, W, ref, bias=b, M=M, tile=256)
    _lib.call("owl_gemm_fr_ablate", 4); chk = torch.zeros_like(o); ops.gemm(epi, A, W, chk, bias=b, M=M, tile=5); torch.cuda.synchronize()
    print("   spread-request variant bits == reference:", bool(torch.equal(chk, ref)))
    for abl, label in ((0, "fr"), (4, "fr requests spread over the MFMAs"), (8, "fr requests never waited for"), (16, "fr, same bytes as WHOLE-line requests"), (20, "whole-line + spread"), (1, "fr no DMA"), (2, "fr no fragment reads"), (3, "fr neither")):
        _lib.call("owl_gemm_fr_ablate", abl)
        res[label] = t(lambda: ops.gemm(epi, A, W, o, bias=b, M=M, tile=5))
    _lib.call("owl_gemm_fr_ablate", 0)
    print(f"{name:4s} N={N} K={K}: " + " | ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
