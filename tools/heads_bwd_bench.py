"""Stand-alone timing of the two head-backward kernels at the headline sizes: box_final_bwd (dense2 backward + erf-GELU derivative of dense1) and merge_ln_bwd
(class-token merge + the two final LayerNorms), each with its fixed-order reductions -- warm and cold (600 MB rewritten between launches), with exact checksums of
every output for same-bits A/B across two builds of libowlhip.so."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import ops, _lib


def csum(t):
    return int(t.contiguous().view(-1).view(torch.uint8).to(torch.int64).mul(torch.arange(t.numel() * t.element_size(), device=t.device) % 251 + 1).sum())


def timed(f, n=20, flush=None):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        if flush is not None: flush.add_(1.0)
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


flush = torch.zeros(150_000_000, device="cuda")
rows, D = 73728, 768
g = torch.Generator(device="cuda").manual_seed(0)
u1 = torch.randn(rows, D, device="cuda", generator=g).bfloat16(); h1 = torch.nn.functional.gelu(u1.float()).bfloat16()
w2 = torch.randn(4, D, device="cuda", generator=g) * 0.1; sig = torch.rand(rows, 4, device="cuda", generator=g); db = torch.randn(rows, 4, device="cuda", generator=g)
du1 = torch.zeros(rows, D, device="cuda", dtype=torch.bfloat16)
part = torch.zeros(_lib.load().owl_box_final_bwd_blocks(rows), 5 * D + 4, device="cuda"); gr = torch.zeros(4 * D + 4, device="cuda")
f = lambda: ops.box_final_bwd(db, sig, h1, u1, w2, du1, part, gr, rows, D)
tw, tc = timed(f), timed(f, flush=flush)
gr.zero_(); f(); torch.cuda.synchronize()
print(f"box_final_bwd: {tw:.1f} us warm / {tc:.1f} cold (with its reductions); checksums du1 {csum(du1)} dW2+db2 {csum(gr)}")

for (B, P, D) in ((32, 2304, 768), (16, 3600, 1024), (1, 2304, 768), (3, 37, 64)):
    T = P + 1; Tp = (T + 7) // 8 * 8
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B * Tp, D, device="cuda", generator=g)
    g1 = torch.randn(D, device="cuda", generator=g); b1 = torch.zeros(D, device="cuda") + 0.1; g2 = torch.randn(D, device="cuda", generator=g)
    b2 = torch.zeros(D, device="cuda")
    cls_ln = torch.zeros(B, D, device="cuda"); feats = ops.zeros_rows(B * P, D, torch.bfloat16, "cuda")
    s1 = torch.zeros(B * Tp, 2, device="cuda"); s2 = torch.zeros(B * P, 2, device="cuda")
    ops.merge_ln(x, g1, b1, g2, b2, cls_ln, feats, s1, s2, B, P, Tp, D)
    df = torch.randn(B * P, D, device="cuda", generator=g)
    dx = torch.zeros(B * Tp, D, device="cuda"); dxb = torch.zeros(B * Tp, D, device="cuda", dtype=torch.bfloat16); dcls = torch.zeros(B, D, device="cuda")
    gr = [torch.zeros(D, device="cuda") for _ in range(4)]; cs = torch.zeros(D, device="cuda")
    part = torch.zeros(B * ((P + 63) // 64) * 64 * D // 8 + 6 * D * B * ((P + 63) // 64), device="cuda")
    f = lambda: ops.merge_ln_bwd(df, x, cls_ln, s1, s2, g1, b1, g2, dx, dcls, *gr, B, P, Tp, D, partials=part, dx_bf16=dxb, dx_colsum=cs)
    tw, tc = timed(f), timed(f, flush=flush)
    for t in gr + [cs]: t.zero_()
    f(); torch.cuda.synchronize()
    print(f"merge_ln_bwd B={B} P={P} D={D}: {tw:.1f} us warm / {tc:.1f} cold (with its reductions); checksums dx {csum(dx)} dx_bf16 {csum(dxb)} feats {csum(feats)} "
          f"params {[csum(t) for t in gr]} colsum {csum(cs)} dcls {csum(dcls)}")
