"""Attention forward with twelve waves per workgroup sharing the K / V stage buffers (variant 5, experimental) against the shipped kernel: bitwise compare + alternating timing."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, T) in ((32, 12, 2305), (8, 12, 2305), (2, 3, 577), (1, 2, 193), (3, 5, 1217)):
    Tp = (T + 7) // 8 * 8; D = H * 64
    torch.manual_seed(T)
    qkv = (torch.randn(B * Tp, 3 * D, device="cuda") * 0.7).bfloat16()
    outs = []
    for v in (0, 5):
        out = torch.zeros(B * Tp, D, device="cuda", dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device="cuda")
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=v)
        outs.append((out, lse))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    out, lse = outs[0]
    res = []
    for rnd in range(3):
        a = t(lambda: ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=0))
        b = t(lambda: ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=5))
        res.append(f"{a:.4f} -> {b:.4f} ms ({(b / a - 1) * 100:+.1f} %)")
    print(f"B={B} H={H} T={T}: bitwise {'same' if same else 'DIFFERENT'}; " + "; ".join(res), flush=True)
