"""Same-process interleaved A/B of the attention-forward sweeps (variant 1 classic, 2 pipelined, 3 pipelined + hints): bit-equality + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def case(B, H, T, rounds=6, iters=20):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    outs = {}
    for v in (1, 2, 3, 4):
        o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
        outs[v] = o
    torch.cuda.synchronize()
    print(f"B={B} H={H} T={T}: bit-equal to classic: pipelined {torch.equal(outs[1], outs[2])}, +hints {torch.equal(outs[1], outs[3])}, optimistic {torch.equal(outs[1], outs[4])}", flush=True)
    o = outs[1]
    times = {1: [], 2: [], 3: [], 4: []}
    for _ in range(3):                                       # warm the clocks
        for v in (1, 2, 3, 4):
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
    for r in range(rounds):
        for v in (1, 2, 3, 4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=v)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / iters)
    fl = 4.0 * B * H * T * T * 64
    for v, name in ((1, "classic        "), (2, "pipelined      "), (3, "pipelined+hints"), (4, "optimistic     ")):
        t = sorted(times[v]); med = t[len(t) // 2]
        print(f"  {name}: median {med:.4f} ms  min {t[0]:.4f} ms  {fl / (med * 1e-3) / 1e12:.0f} TF/s", flush=True)


if __name__ == "__main__":
    case(32, 12, 2305)
    case(16, 16, 3601)
    case(8, 12, 577)
