"""Time the attention backward launches (dvec + dK/dV + dQ) at B/16 batch-32 and L/14 batch-16 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
def run(B, H, T, iters=10):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    qkvT = torch.zeros(B * 3 * D * Tp + 256, device=DEV, dtype=torch.bfloat16)
    qkvT[: B * 3 * D * Tp].view(B, 3 * D, Tp)[:] = qkv[:M].view(B, Tp, 3 * D).transpose(1, 2)
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125)
    do = torch.zeros_like(o); do[:M] = (torch.randn(M, D, device=DEV) * 0.1).bfloat16()
    doT = torch.zeros(B * D * Tp + 256, device=DEV, dtype=torch.bfloat16); doT[: B * D * Tp].view(B, D, Tp)[:] = do[:M].view(B, Tp, D).transpose(1, 2)
    dvec = torch.zeros(B, H, Tp, device=DEV); dqkv = torch.zeros_like(qkv)
    f = lambda: _lib.call("owl_attention_bwd_bf16", ops.stream(), qkv, do, o, lse, dvec, dqkv, B, H, T, Tp, 0.125, 0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters
    print(f"attention bwd B={B} H={H} T={T}: {t:.3f} ms  ({10.0 * B * H * T * T * 64 / t / 1e9:.0f} TF/s on the minimal 5 matmuls)", flush=True)
for _ in range(3): run(32, 12, 2305)
run(16, 16, 3601)
