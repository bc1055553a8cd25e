"""Reference point, not product: the vendor GEMM (hipBLASLt through torch.nn.functional.linear) at the hot path's
GEMM shapes, interleaved with this repo's kernel, to see how much head-room the hand-written kernel leaves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
from microbench import timeit

DEV = "cuda"
B = 32; Tp = 2312; M = B * Tp
shapes = [(M, 768, 768), (M, 2304, 768), (M, 3072, 768), (M, 768, 3072), (8192, 8192, 8192)]
for _ in range(20):
    a = torch.randn(8192, 8192, device=DEV).bfloat16(); a @ a
for (m, n, k) in shapes:
    A = torch.randn(ops.pad_rows(m), k, device=DEV).bfloat16()
    W = (torch.randn(n, k, device=DEV) * 0.05).bfloat16()
    bias = torch.randn(n, device=DEV)
    biasb = bias.bfloat16()
    out = torch.zeros(ops.pad_rows(m), n, device=DEV, dtype=torch.bfloat16)
    for rep in range(2):
        t1 = timeit(lambda: ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=m))
        t2 = timeit(lambda: torch.nn.functional.linear(A[:m], W, biasb))
        t3 = timeit(lambda: torch.matmul(A[:m], W.t()))
        fl = 2.0 * m * n * k / 1e12
        print(f"M={m} N={n} K={k}: ours {t1*1e3:.3f} ms {fl/t1:.0f} TF/s | linear+bias {t2*1e3:.3f} ms {fl/t2:.0f} TF/s | "
              f"matmul {t3*1e3:.3f} ms {fl/t3:.0f} TF/s", flush=True)
