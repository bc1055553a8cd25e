import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"
def run(rows, n_out, n_in, splits, fill):
    dy = torch.zeros(ops.pad_rows(rows), n_out, device=DEV, dtype=torch.bfloat16)
    m = torch.arange(rows, device=DEV)[:, None]; n = torch.arange(n_out, device=DEV)[None, :]
    dy[:rows] = fill(m, n).bfloat16()
    x = torch.zeros(ops.pad_rows(rows), n_in, device=DEV, dtype=torch.bfloat16); x[:rows] = 1
    slab = torch.zeros(splits * n_out * n_in, device=DEV); bslab = torch.zeros(splits, n_out, device=DEV)
    ns = ops.gemm_tn_slab(dy, x, slab, rows, n_out, n_in, splits, bias_slab=bslab)
    got = bslab[:ns].sum(0); ref = dy[:rows].float().sum(0)
    w = slab[: ns * n_out * n_in].view(ns, n_out, n_in).sum(0)[:, 0]      # = colsum too (x = ones)
    print(f"rows {rows} n_out {n_out} n_in {n_in} ns {ns}: bias-slab max err {float((got-ref).abs().max()):.3f}, weight-slab col 0 max err {float((w-ref).abs().max()):.3f}")
    print("   got", got[:8].tolist(), "\n   ref", ref[:8].tolist(), "\n   got[32:40]", got[32:40].tolist(), "ref", ref[32:40].tolist(), "\n   got[128:132]", got[128:132].tolist(), "ref", ref[128:132].tolist(), "\n   got[256:260]", got[256:260].tolist(), "ref", ref[256:260].tolist())
    print("   per split row 0:", bslab[:ns, 0].tolist()[:6])
ones = lambda m, n: torch.ones(m.shape[0], n.shape[1], device=DEV)
run(64, 256, 256, 1, ones)
run(64, 256, 256, 1, lambda m, n: (n % 7).float().expand(m.shape[0], -1))
run(64, 256, 256, 1, lambda m, n: ((m % 16) < 8).float().expand(-1, n.shape[1]))
run(64, 256, 256, 1, lambda m, n: (m.float() / 64).expand(-1, n.shape[1]))
run(128, 256, 256, 1, ones)
run(128, 256, 256, 2, ones)
run(640, 512, 768, 3, lambda m, n: (n % 5).float().expand(m.shape[0], -1))
