"""dW = dY^T X on the TN kernels: single-phase (variant 1) vs ping-pong (variant 2), same process, interleaved; bits + time at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"
for M, shapes in ((32 * 2312, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]), (32 * 2304, [(768, 768), (512, 768)]), (16 * 3608, [(4096, 1024), (1024, 4096)])):
    for (n_out, n_in) in shapes:
        dy = torch.zeros(ops.pad_rows(M), n_out, device=DEV, dtype=torch.bfloat16); dy[:M] = (torch.randn(M, n_out, device=DEV) * 0.1).bfloat16()
        x = torch.zeros(ops.pad_rows(M), n_in, device=DEV, dtype=torch.bfloat16); x[:M] = torch.randn(M, n_in, device=DEV).bfloat16()
        tiles = (n_out // 256) * (n_in // 256); splits = max(1, 256 // tiles)
        slabs = {v: torch.zeros(splits * n_out * n_in, device=DEV) for v in (1, 2)}
        ns = {v: ops.gemm_tn_slab(dy, x, slabs[v], M, n_out, n_in, splits, variant=v) for v in (1, 2)}
        torch.cuda.synchronize()
        eq = ns[1] == ns[2] and torch.equal(slabs[1], slabs[2])
        times = {1: [], 2: []}
        for _ in range(2):
            for v in (1, 2):
                for _ in range(5): ops.gemm_tn_slab(dy, x, slabs[v], M, n_out, n_in, splits, variant=v)
        for r in range(5):
            for v in (1, 2):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                for _ in range(10): ops.gemm_tn_slab(dy, x, slabs[v], M, n_out, n_in, splits, variant=v)
                e1.record(); torch.cuda.synchronize(); times[v].append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * M * n_out * n_in
        t1, t2 = sorted(times[1])[2], sorted(times[2])[2]
        print(f"dW [{n_out} x {n_in}] over {M} rows, {ns[1]} splits: bits equal {eq};  single-phase {t1*1e3:7.1f} us {fl/t1/1e9:5.0f} TF/s | ping-pong {t2*1e3:7.1f} us {fl/t2/1e9:5.0f} TF/s  ({(t1/t2-1)*100:+.1f} %)", flush=True)
