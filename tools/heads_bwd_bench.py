"""Stand-alone timing of the two head-backward kernels at the headline sizes: box_final_bwd (dense2 backward + erf-GELU derivative of dense1)
and merge_ln_bwd (class-token merge + the two final LayerNorms), each with its fixed-order reductions.  Prints a checksum for same-bits A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import ops, _lib
rows, D = 73728, 768
g = torch.Generator(device="cuda").manual_seed(0)
u1 = torch.randn(rows, D, device="cuda", generator=g).bfloat16(); h1 = torch.nn.functional.gelu(u1.float()).bfloat16()
w2 = torch.randn(4, D, device="cuda", generator=g) * 0.1; sig = torch.rand(rows, 4, device="cuda", generator=g); db = torch.randn(rows, 4, device="cuda", generator=g)
du1 = torch.zeros(rows, D, device="cuda", dtype=torch.bfloat16)
part = torch.zeros(_lib.load().owl_box_final_bwd_blocks(rows), 5 * D + 4, device="cuda"); gr = torch.zeros(4 * D + 4, device="cuda")
for _ in range(5): ops.box_final_bwd(db, sig, h1, u1, w2, du1, part, gr, rows, D)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(50): ops.box_final_bwd(db, sig, h1, u1, w2, du1, part, gr, rows, D)
e1.record(); torch.cuda.synchronize()
print("box_final_bwd us", e0.elapsed_time(e1) / 50 * 1e3, "checksum", float(du1.float().sum()), float(gr.sum()))

for (B, P, D) in ((32, 2304, 768), (16, 3600, 1024)):
    T = P + 1; Tp = (T + 7) // 8 * 8
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B * Tp, D, device="cuda", generator=g)
    g1 = torch.ones(D, device="cuda"); b1 = torch.zeros(D, device="cuda") + 0.1; g2 = torch.ones(D, device="cuda")
    b2 = torch.zeros(D, device="cuda")
    cls_ln = torch.zeros(B, D, device="cuda"); feats = ops.zeros_rows(B * P, D, torch.bfloat16, "cuda")
    s1 = torch.zeros(B * Tp, 2, device="cuda"); s2 = torch.zeros(B * P, 2, device="cuda")
    ops.merge_ln(x, g1, b1, g2, b2, cls_ln, feats, s1, s2, B, P, Tp, D)
    df = torch.randn(B * P, D, device="cuda", generator=g)
    dx = torch.zeros(B * Tp, D, device="cuda"); dxb = torch.zeros(B * Tp, D, device="cuda", dtype=torch.bfloat16); dcls = torch.zeros(B, D, device="cuda")
    gr = [torch.zeros(D, device="cuda") for _ in range(4)]
    part = torch.zeros(B * ((P + 63) // 64) * 64 * D // 8 + 5 * D * B * ((P + 63) // 64), device="cuda")
    f = lambda: ops.merge_ln_bwd(df, x, cls_ln, s1, s2, g1, b1, g2, dx, dcls, *gr, B, P, Tp, D, partials=part, dx_bf16=dxb)
    for _ in range(5): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"merge_ln_bwd B={B} P={P} D={D}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (with its reductions)  checksum {float(dx.double().sum()):.6e} {float(gr[0].sum()):.6e}")
