"""Same-process A/B of the GEMM dispatch at the model's shapes: tile=8 (256x256 ping-pong kernel for every tile) against tile=0
(automatic: whole rounds on the 256x256 kernel + the remainder rows on the half-height 128x256 variant where that pays): bits + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def case(M, N, K, epi, rounds=6, iters=20):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    outs = {}
    for t in (8, 9):
        o = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(epi, A, W, o, bias=bias, M=M, tile=t)
        outs[t] = o
    torch.cuda.synchronize()
    o = outs[8]
    times = {8: [], 9: []}
    for _ in range(2):
        for t in (8, 9):
            for _ in range(iters): ops.gemm(epi, A, W, o, bias=bias, M=M, tile=t)
    for r in range(rounds):
        for t in (8, 9):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.gemm(epi, A, W, o, bias=bias, M=M, tile=t)
            e1.record(); torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / iters)
    fl = 2.0 * M * N * K
    m8 = sorted(times[8])[len(times[8]) // 2]; m0 = sorted(times[9])[len(times[9]) // 2]
    print(f"M={M} N={N} K={K} epi={epi}: equal={torch.equal(outs[8], outs[9])}  pp-only {m8:.4f} ms ({fl/m8/1e9:.0f} TF/s)  auto {m0:.4f} ms ({fl/m0/1e9:.0f} TF/s)  {100*(m8/m0-1):+.1f} %", flush=True)


if __name__ == "__main__":
    M = 32 * 2312
    with torch.no_grad():
        for _ in range(30):   # warm the clocks
            ops.gemm(ops.EPI_BIAS_BF16, torch.zeros(8192, 8192, device=DEV, dtype=torch.bfloat16), torch.zeros(8192, 8192, device=DEV, dtype=torch.bfloat16),
                     torch.zeros(8192, 8192, device=DEV, dtype=torch.bfloat16), M=8192) if _ == 0 else None
    case(M, 768, 768, ops.EPI_BIAS_BF16)
    case(M, 2304, 768, ops.EPI_BIAS_BF16)
    case(M, 3072, 768, ops.EPI_QGELU_BF16)
    case(M, 768, 3072, ops.EPI_BIAS_BF16)
    case(M, 768, 2304, ops.EPI_BIAS_BF16)
    case(32 * 2304, 768, 768, ops.EPI_GELU_BF16)
    case(16 * 3608, 4096, 1024, ops.EPI_QGELU_BF16)
    case(16 * 3608, 1024, 4096, ops.EPI_BIAS_BF16)
