import sys, os
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops
DEV="cuda"
for (M,N,K) in ((73984,2304,768),(73984,768,3072),(73984,768,768),(73984,3072,768)):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
    epi = ops.EPI_QGELU_BF16 if N == 3072 else ops.EPI_BIAS_BF16
    for _ in range(10): ops.gemm(epi, A, W, out, bias=bias, M=M)
    torch.cuda.synchronize()
    res = []
    for n in (1, 2, 5, 20, 50):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): ops.gemm(epi, A, W, out, bias=bias, M=M)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        res.append(f"{n}: {sorted(ts)[2]:.1f}")
    print(f"M={M} N={N} K={K}: us per call by number of back-to-back calls -> " + ", ".join(res), flush=True)
