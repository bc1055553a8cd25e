import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
B, H, T = 32, 12, 2305
Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
def t(v, iters=20):
    f = lambda: ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
    for _ in range(10): f()
    ts = []
    for r in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[2]
lib = _lib.load()
for flags, name in ((0, "full"), (256, "groups = w & 1"), (512, "groups = (w >> 1) & 1"), (256 | 8, "groups = w & 1, no softmax"), (256 | 16, "groups = w&1, no M"), (32, "no stagger (both groups in the same phase)"), (8, "no softmax"), (16, "no M phase"), (8 | 16, "no softmax, no M phase (reads + DMA + barriers)"),
                    (64, "no DMA"), (8 | 16 | 64, "K reads + barriers only"), (0, "full")):
    lib.owl_attention_debug(flags)
    print(f"{name:55s}: {t(3):.4f} ms   (free-running peeled: {t(2):.4f})", flush=True)
lib.owl_attention_debug(0)
