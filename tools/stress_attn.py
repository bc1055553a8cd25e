"""Race screen of the attention kernels: repeated launches under concurrent HBM traffic must be bitwise identical
(forward output + lse; backward dQ/dK/dV), at the B/16 batch-32 and L/14 batch-4 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
torch.manual_seed(0)
bad = 0
big = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.uint8)
s2 = torch.cuda.Stream()
for (B, H, T, iters) in [(32, 12, 2305, 100), (4, 16, 3601, 60), (3, 12, 577, 100)]:
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    qkvT = torch.zeros(B * 3 * D * Tp + 256, device=DEV, dtype=torch.bfloat16)
    qkvT[: B * 3 * D * Tp].view(B, 3 * D, Tp)[:] = qkv[:M].view(B, Tp, 3 * D).transpose(1, 2)
    do = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); do[:M] = (torch.randn(M, D, device=DEV) * 0.1).bfloat16()
    doT = torch.zeros(B * D * Tp + 256, device=DEV, dtype=torch.bfloat16); doT[: B * D * Tp].view(B, D, Tp)[:] = do[:M].view(B, Tp, D).transpose(1, 2)

    def run():
        o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device=DEV)
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125)      # the variant the model runs
        dvec = torch.zeros(B, H, Tp, device=DEV); dqkv = torch.zeros_like(qkv)
        _lib.call("owl_attention_bwd_bf16", ops.stream(), qkv, do, o, lse, dvec, dqkv, B, H, T, Tp, 0.125, 0)
        return o, lse, dqkv

    ref = run()
    for it in range(iters):
        with torch.cuda.stream(s2):
            big.add_(1)
        got = run()
        for a, b, name in zip(got, ref, ("out", "lse", "dqkv")):
            if not torch.equal(a, b):
                bad += 1
                print("MISMATCH", B, H, T, it, name, (a.float() - b.float()).abs().max().item(), flush=True)
    torch.cuda.synchronize()
    print("shape", B, H, T, "done", flush=True)
print("mismatches:", bad)
