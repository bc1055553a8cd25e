import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
B, H, T = 32, 12, 2305
Tp = (T + 7)//8*8; D = H*64; M = B*Tp
qkv = torch.randn(ops.pad_rows(M), 3*D, device="cuda").bfloat16()
vt = torch.randn(B*H*64*Tp + 128, device="cuda").bfloat16()
out = torch.zeros(ops.pad_rows(M), D, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention_fwd(qkv, qkv[:, D:], 3*D, vt, H*64*Tp, out, D, None, B, H, T, Tp, 0.125)
torch.cuda.synchronize()
