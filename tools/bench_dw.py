"""dW = dY^T X: TN kernel (in-place operands) vs transposes + NT slab kernel, at the B/16 batch-32 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M = 73984
for _ in range(2):
  for (n_out, n_in) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    dy = (torch.randn(M, n_out, device=DEV) * 0.1).bfloat16(); x = torch.randn(M, n_in, device=DEV).bfloat16()
    tiles = (n_out // 256) * (n_in // 256); splits = max(1, 256 // tiles)
    slab = torch.zeros(splits * n_out * n_in, device=DEV)
    t_tn = timeit(lambda: ops.gemm_tn_slab(dy, x, slab, M, n_out, n_in, splits))
    tA = torch.zeros(n_out, M, device=DEV, dtype=torch.bfloat16); tB = torch.zeros(n_in, M, device=DEV, dtype=torch.bfloat16)
    cs = torch.zeros(n_out, device=DEV)
    def old():
        ops.transpose_colsum(dy, tA, cs, M, n_out, ld_in=n_out, ld_out=M)
        ops.transpose_colsum(x, tB, None, M, n_in, ld_in=n_in, ld_out=M)
        ops.gemm(ops.EPI_SLAB_F32, tA, tB, slab, M=n_out, N=n_in, K=M, lda=M, ldw=M, ldo=n_in, a_rows=n_out, w_rows=n_in, splits=splits)
    t_old = timeit(old)
    t_cs = timeit(lambda: ops.colsum_bf16(dy, cs, M, n_out))
    fl = 2.0 * M * n_out * n_in
    print(f"dW {n_out}x{n_in}: TN {t_tn:.3f} ms ({fl/t_tn/1e9:.0f} TF/s) + colsum {t_cs:.3f} | transposes+NT {t_old:.3f} ms", flush=True)
