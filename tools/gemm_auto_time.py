"""Time of the automatic GEMM choice (tile 0) at the model's narrow-output shapes + bits against the single-phase reference kernel (same-box A/B of two library builds: tools/ab_round4.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
M = 32 * 2312
for name, N, K, epi in (("out-proj", 768, 768, ops.EPI_BIAS_BF16), ("fc2", 768, 3072, ops.EPI_BIAS_BF16), ("dX K=2304", 768, 2304, ops.EPI_BIAS_BF16), ("QKV", 2304, 768, ops.EPI_BIAS_BF16), ("fc1", 3072, 768, ops.EPI_QGELU_BF16)):
    torch.manual_seed(1)
    A = torch.randn(ops.pad_rows(M), K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    o = torch.zeros(ops.pad_rows(M), N, device="cuda", dtype=torch.bfloat16); ref = torch.zeros_like(o)
    ops.gemm(epi, A, W, ref, bias=b, M=M, tile=256); ops.gemm(epi, A, W, o, bias=b, M=M, tile=0); torch.cuda.synchronize()
    same = torch.equal(o, ref)
    for _ in range(30): ops.gemm(epi, A, W, o, bias=b, M=M)
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(20): ops.gemm(epi, A, W, o, bias=b, M=M)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{name:10s} N={N} K={K}: {sorted(ts)[2]:7.1f} us  bits == reference: {same}", flush=True)
