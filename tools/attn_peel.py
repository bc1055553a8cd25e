"""Attention forward: plain tiling (variant 1) vs class token peeled (variant 2), interleaved in one process -- time, max deviation from each
other and from f32 softmax.  Also the plain tiling at T - 1 (the bound on what peeling can win)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def case(B, H, T, rounds=7, iters=20):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    outs = {}
    for v in (1, 2):
        o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
        outs[v] = o
    x = qkv[:M].view(B, Tp, 3, H, 64)[:1, :T].float()
    q, k, vv = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    want = (torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ vv).permute(0, 2, 1, 3).reshape(T, D)
    e = {v: (outs[v][:Tp][:T].float() - want).abs().max().item() for v in (1, 2)}
    print(f"B={B} H={H} T={T}: max|plain - peeled| {(outs[1].float() - outs[2].float()).abs().max().item():.2e}; vs f32 softmax (image 0): plain {e[1]:.2e}, peeled {e[2]:.2e}", flush=True)
    o = outs[1]
    runs = {"plain": (T, 1), "peeled": (T, 2), "plain, T-1": (T - 1, 1)}
    times = {n: [] for n in runs}
    for _ in range(3):
        for n, (t, v) in runs.items():
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, t, Tp, 0.125, variant=v)
    for r in range(rounds):
        for n, (t, v) in runs.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, t, Tp, 0.125, variant=v)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / iters)
    fl = 4.0 * B * H * T * T * 64
    for n in runs:
        t = sorted(times[n]); med = t[len(t) // 2]
        print(f"  {n:11s}: median {med:.4f} ms  min {t[0]:.4f} ms  {fl / (med * 1e-3) / 1e12:.0f} TF/s (T = {T} flops)", flush=True)


if __name__ == "__main__":
    case(32, 12, 2305)
    case(32, 12, 577)
    case(16, 16, 3585)
