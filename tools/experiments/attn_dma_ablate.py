"""OWL_TUNING build, timing only (the ablated launch computes garbage): the shipped attention forward with every LDS-DMA piece against ONE of a wave's four
pieces from the third tile on -- what a workgroup of 12-16 waves sharing the K / V stage buffers would issue and fetch.  B/16 shape."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops, _lib
B, H, T = 32, 12, 2305
Tp = 2312
D = H * 64
torch.manual_seed(0)
qkv = (torch.randn(B * Tp, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.zeros(B * Tp, D, device="cuda", dtype=torch.bfloat16)
def run():
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125)
def t(n=20):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(3):
    _lib.call("owl_attention_debug", 0); a = t()
    _lib.call("owl_attention_debug", 8); b = t()
    print(f"all DMA pieces {a:.4f} ms; one of four pieces from tile 2 on {b:.4f} ms ({(b / a - 1) * 100:+.1f} %)", flush=True)
_lib.call("owl_attention_debug", 0)
