import os, sys
sys.path.insert(0, os.getcwd())
import torch
from owl_vit_object_detection_amd import ops
for name, M, N, K in (("B/16 dX through quick-GELU'", 32 * 2312, 3072, 768), ("L/14 dX through quick-GELU'", 16 * 3608, 4096, 1024)):
    torch.manual_seed(1)
    A = torch.randn(ops.pad_rows(M), K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    aux = torch.randn(ops.pad_rows(M), N, device="cuda").bfloat16()
    o = torch.zeros(ops.pad_rows(M), N, device="cuda", dtype=torch.bfloat16); ref = torch.zeros_like(o)
    ops.gemm(ops.EPI_DQGELU_BF16, A, W, ref, aux=aux, M=M, tile=256); ops.gemm(ops.EPI_DQGELU_BF16, A, W, o, aux=aux, M=M); torch.cuda.synchronize()
    same = torch.equal(o, ref)
    for _ in range(20): ops.gemm(ops.EPI_DQGELU_BF16, A, W, o, aux=aux, M=M)
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): ops.gemm(ops.EPI_DQGELU_BF16, A, W, o, aux=aux, M=M)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{name}: {sorted(ts)[2]:7.1f} us  bits == reference: {same}", flush=True)
