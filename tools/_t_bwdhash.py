import sys, os, hashlib
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
for (B, H, T) in [(2, 3, 333), (1, 2, 37), (2, 12, 2305), (3, 12, 577)]:
    torch.manual_seed(T)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    qkvT = torch.zeros(B * 3 * D * Tp + 256, device=DEV, dtype=torch.bfloat16)
    qkvT[: B * 3 * D * Tp].view(B, 3 * D, Tp)[:] = qkv[:M].view(B, Tp, 3 * D).transpose(1, 2)
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125)
    do = torch.zeros_like(o); do[:M] = (torch.randn(M, D, device=DEV) * 0.1).bfloat16()
    doT = torch.zeros(B * D * Tp + 256, device=DEV, dtype=torch.bfloat16); doT[: B * D * Tp].view(B, D, Tp)[:] = do[:M].view(B, Tp, D).transpose(1, 2)
    dvec = torch.zeros(B, H, Tp, device=DEV); dqkv = torch.zeros_like(qkv)
    _lib.call("owl_attention_bwd_bf16", ops.stream(), qkv, qkvT, do, doT, o, lse, dvec, dqkv, B, H, T, Tp, 0.125)
    torch.cuda.synchronize()
    hsh = hashlib.md5(dqkv[:M].view(torch.int16).cpu().numpy().tobytes()).hexdigest()
    print(B, H, T, hsh, bool(torch.isfinite(dqkv.float()).all()))
