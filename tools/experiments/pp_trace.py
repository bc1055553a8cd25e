"""Half-slot timeline of the ping-pong GEMM (tuning only): workgroup 0 stamps s_memtime at every LOAD/MFMA boundary of
K-tile 4 of its first tile (gemm_pp_kernel<BIAS, TRACE>); prints per-wave segment lengths in cycles.

stamp index inside a K-tile (quadrant q = 0..3, 5 stamps each):
  5q+0 LOAD half starts | 5q+1 loads issued | 5q+2 lgkmcnt/vmcnt wait done | 5q+3 barrier passed (MFMA half starts)
  | 5q+4 MFMAs issued (then the closing barrier -> next 5(q+1)+0)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops, _lib

DEV = "cuda"
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8192)))
A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
bias = torch.randn(N, device=DEV)
out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
trace = torch.zeros(320, dtype=torch.int64, device=DEV)
ops.GEMM_TILE = 8
for _ in range(5):
    ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M)
_lib.call("owl_gemm_debug_nostore", 8)
ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M, aux=trace, ld_aux=1)
torch.cuda.synchronize()
_lib.call("owl_gemm_debug_nostore", 0)
t = trace.cpu().numpy().reshape(8, 40)
t0 = t.min()
names = ["issue", "wait", "bar", "mfma", "bar"]
print(f"M={M} N={N} K={K}; cycles relative to the first stamp; one row per wave (group = wave>>2)")
for w in range(8):
    d = t[w] - t0
    segs = []
    for i in range(19):
        segs.append(int(d[i + 1] - d[i]))
    line = " | ".join(" ".join(f"{segs[q * 5 + j]:4d}" for j in range(5) if q * 5 + j < 19) for q in range(4))
    print(f"w{w} start {int(d[0]):5d}: {line}")
print("columns per quadrant: " + " ".join(names) + "   (issue = ds_read/DMA issue, wait = s_waitcnt, bar = barrier before MFMAs, mfma = 8 MFMAs issued, bar = closing barrier)")
tot = (t[:, 19] - t[:, 0]).mean() * 20 / 19
print(f"mean cycles per K-tile per wave ~ {tot:.0f} (ideal 2048: 64 MFMA x 32 per SIMD)")
