"""Reference point, not product: the vendor's fused attention (torch.nn.functional.scaled_dot_product_attention on ROCm -- CK / AOTriton flash
kernels) at the hot path's attention shape, beside this repo's kernel.  B = 32, H = 12, T = 2305, dh = 64, bf16, no mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def timeit(fn, n=20, rounds=5):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]


for (B, H, T) in ((32, 12, 2305), (32, 12, 2304), (16, 16, 3601)):
    fl = 4.0 * B * H * T * T * 64
    q, k, v = (torch.randn(B, H, T, 64, device=DEV).bfloat16() for _ in range(3))
    res = []
    for name, backend in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION), ("math", torch.nn.attention.SDPBackend.MATH)):
        try:
            with torch.nn.attention.sdpa_kernel(backend):
                t = timeit(lambda: F.scaled_dot_product_attention(q, k, v), n=10 if name != "math" else 2, rounds=3)
            res.append(f"SDPA {name} {t:.3f} ms = {fl / t / 1e9:.0f} TF/s")
        except Exception as e:
            res.append(f"SDPA {name}: unavailable ({type(e).__name__})")
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda: ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125))
    print(f"B={B} H={H} T={T}: " + "; ".join(res) + f"; this repo {t:.3f} ms = {fl / t / 1e9:.0f} TF/s", flush=True)
