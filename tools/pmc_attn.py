"""Launch the attention-forward variants a few times each (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"
B, H, T = 32, 12, 2305
Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
qkv = torch.randn(ops.pad_rows(M), 3 * D, device=DEV).bfloat16()
out = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
variants = [int(v) for v in (sys.argv[1:] or ["1", "2"])]
for v in variants:
    for _ in range(5):
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, variant=v)
torch.cuda.synchronize()
