"""OWL_TUNING build: the two-phase ping-pong GEMM with and without its epilogue stores (upper bound on the cost of the store path)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV="cuda"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M,N,K) in ((73984,2304,768),(73984,768,768),(73984,768,3072),(73984,3072,768)):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
    epi = ops.EPI_QGELU_BF16 if N == 3072 else ops.EPI_BIAS_BF16
    ops.GEMM_TILE = 7
    res = []
    for rep in range(2):
        _lib.call("owl_gemm_pp2_nostore", 0); a = t(lambda: ops.gemm(epi, A, W, out, bias=bias, M=M))
        _lib.call("owl_gemm_pp2_nostore", 1); b = t(lambda: ops.gemm(epi, A, W, out, bias=bias, M=M))
        res.append(f"stores {a:.1f} us / no stores {b:.1f} us")
    _lib.call("owl_gemm_pp2_nostore", 0)
    print(f"M={M} N={N} K={K} (two-phase kernel on the whole problem): " + "; ".join(res), flush=True)
