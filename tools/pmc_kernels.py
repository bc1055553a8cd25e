"""Launch the two dominant kernels a few times each (for rocprofv3 --pmc passes): attention forward and the fc1 GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"
B, H, T = 32, 12, 2305
Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
qkv = torch.randn(ops.pad_rows(M), 3 * D, device=DEV).bfloat16()
vt = torch.randn(B * H * 64 * Tp + 128, device=DEV).bfloat16()
out = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
A = torch.randn(ops.pad_rows(M), 768, device=DEV).bfloat16()
W = (torch.randn(3072, 768, device=DEV) * 0.05).bfloat16(); bias = torch.randn(3072, device=DEV)
u = torch.zeros(ops.pad_rows(M), 3072, device=DEV, dtype=torch.bfloat16)
for _ in range(5):
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125)     # the variant the model runs
    ops.gemm(ops.EPI_QGELU_BF16, A, W, u, bias=bias, M=M)
torch.cuda.synchronize()
