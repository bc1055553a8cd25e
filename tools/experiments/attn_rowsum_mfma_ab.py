"""Round-5 timing experiment (NOT in the repo's kernel: tools/probe/attn_rowsum_mfma.patch on csrc/attention_fwd.hip, OWL_TUNING build): the attention forward's 32 VALU
adds per 64-key tile (row sums of P) moved to the matrix pipe -- one extra MFMA per 16 keys with an all-ones A operand -- and the overflow verdict dropped (timing only:
how much VALU relief is worth before a replacement verdict is designed).  owl_attention_debug(16) switches it on; 0 = the shipped behaviour inside the same library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import _lib, ops
assert _lib.is_tuning_build()
for B, H, T in ((32, 12, 2305), (16, 16, 3601)):
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    torch.manual_seed(1)
    qkv = torch.randn(ops.pad_rows(M), 3 * D, device="cuda").bfloat16()
    outs, ts = {}, {0: [], 16: []}
    for rnd in range(4):
        for dbg in (0, 16):
            _lib.call("owl_attention_debug", dbg)
            out = torch.zeros(ops.pad_rows(M), D, device="cuda", dtype=torch.bfloat16)
            for _ in range(5): ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125)
            e1.record(); torch.cuda.synchronize(); ts[dbg].append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[dbg] = out
    _lib.call("owl_attention_debug", 0)
    err = float((outs[16].float() - outs[0].float()).abs().max())
    a, b = sorted(ts[0])[1], sorted(ts[16])[1]
    print(f"B={B} H={H} T={T}: shipped path {a:.1f} us, row sums on the matrix pipe {b:.1f} us ({(b / a - 1) * 100:+.1f} %), max |out diff| {err:.2e}", flush=True)
