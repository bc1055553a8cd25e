"""Attention forward: classic peeled kernel (variant 2: 3 waves per SIMD, 32 queries per wave) vs the one-wave-per-SIMD kernel (variant 3:
64 queries per wave), interleaved in one process -- max deviation from each other and from f32 softmax (output + LSE), the redo path on spiked
scores, time."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def ref(qkv, B, H, T, Tp, nb=1):
    D = H * 64
    v = qkv[: B * Tp].view(B, Tp, 3, H, 64)[:nb, :T].float()
    q, k, vv = (v[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(2, 3) * 0.125
    return (torch.softmax(sc, -1) @ vv).permute(0, 2, 1, 3).reshape(nb, T, D), torch.logsumexp(sc, -1) / math.log(2.0)


def run(qkv, B, H, T, Tp, variant, lse=True):
    D = H * 64; M = B * Tp
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
    l = torch.zeros(B, H, Tp, device=DEV) if lse else None
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, l, B, H, T, Tp, 0.125, variant=variant)
    return o, l


def check(B, H, T, spike=None):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    torch.manual_seed(T + B)
    x = torch.randn(B, T, 3 * D, device=DEV)
    if spike is not None:
        key, qq, gain = spike
        x[0, key, D:D + 64] = gain * torch.sign(x[0, qq, :64] + 1e-3)
        x[0, qq, :64] *= 6.0
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = x.bfloat16()
    nb = min(B, 2)
    want, lse_want = ref(qkv, B, H, T, Tp, nb)
    o2, l2 = run(qkv, B, H, T, Tp, 2)
    o3, l3 = run(qkv, B, H, T, Tp, 3)
    torch.cuda.synchronize()
    redo = int(ops.attention_redo_ws(B, H, T, DEV)[: B * H * ((T - 1 + 255) // 256)].sum())
    v = lambda o: o[:M].view(B, Tp, D)[:nb, :T].float()
    e2, e3 = (v(o2) - want).abs().max().item(), (v(o3) - want).abs().max().item()
    le2, le3 = (l2[:nb, :, :T] - lse_want).abs().max().item(), (l3[:nb, :, :T] - lse_want).abs().max().item()
    o3b, _ = run(qkv, B, H, T, Tp, 3)
    print(f"B={B} H={H} T={T} spike={spike}: max|v3 - v2| {(o3.float() - o2.float()).abs().max().item():.2e}; vs f32: out v2 {e2:.2e} v3 {e3:.2e}; "
          f"lse v2 {le2:.2e} v3 {le3:.2e}; blocks redone {redo}; repeatable {bool(torch.equal(o3, o3b))}; finite {bool(torch.isfinite(o3.float()).all())}", flush=True)


def bench(B, H, T, rounds=7, iters=20):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
    runs = {"classic (v2)": 2, "one wave/SIMD (v3)": 3}
    times = {n: [] for n in runs}
    for _ in range(3):
        for n, v in runs.items():
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
    for r in range(rounds):
        for n, v in runs.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / iters)
    fl = 4.0 * B * H * T * T * 64
    print(f"B={B} H={H} T={T}")
    for n in runs:
        t = sorted(times[n]); med = t[len(t) // 2]
        print(f"  {n:20s}: median {med:.4f} ms  min {t[0]:.4f} ms  {fl / (med * 1e-3) / 1e12:.0f} TF/s", flush=True)


if __name__ == "__main__":
    if "--bench-only" not in sys.argv:
        check(1, 2, 2305)
        check(2, 3, 577)
        check(1, 1, 193)
        check(1, 1, 257)
        check(3, 2, 1025)
        check(1, 2, 2305, spike=(250, 10, 12.0))
        check(1, 2, 2305, spike=(0, 5, 12.0))
        check(1, 2, 2305, spike=(2304, 2000, 12.0))
        check(4, 12, 2305)
    bench(32, 12, 2305)
    bench(16, 12, 2305)
