import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
B, H, T = 32, 12, 2305
Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
lse = torch.zeros(B, H, Tp, device=DEV)
lib = _lib.load()
names = ["V start", "softmax done", "K issued", "M start (after barrier)", "K waited", "tr issued", "QK issued", "DMA issued", "PV first pair issued", "PV issued", "M end", "after barrier"]
for flags in (1024, 1024 | 8):
    lib.owl_attention_debug(flags)
    for _ in range(3):
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125, variant=3)
    torch.cuda.synchronize()
    tr = o[T].view(torch.int32)[:32].cpu().view(2, 16)
    print("flags", flags)
    for g in range(2):
        t = tr[g, :12].tolist()
        print(f" group {g}: " + "  ".join(f"{names[i]}: +{(t[i] - t[0]) & 0xffffffff}" for i in range(12)))
    print(f" group1 V start - group0 V start: {(tr[1,0]-tr[0,0]).item()}")
lib.owl_attention_debug(0)
