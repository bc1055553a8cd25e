"""box_final (dense2 + box bias + sigmoid + corners) and the image cast at the headline sizes: cold time per launch + exact checksums for same-bits A/B across two builds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import ops


def csum(t):
    return int(t.contiguous().view(-1).view(torch.uint8).to(torch.int64).mul(torch.arange(t.numel() * t.element_size(), device=t.device) % 251 + 1).sum())


def timed(f, n=30, flush=None):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        if flush is not None: flush.add_(1.0)
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


flush = torch.zeros(150_000_000, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for B, P, D in ((32, 2304, 768), (16, 3600, 1024), (1, 2304, 768), (3, 37, 128), (2, 36, 520)):
    rows = B * P
    h = torch.randn(rows, D, device="cuda", generator=g).bfloat16(); w2 = torch.randn(4, D, device="cuda", generator=g) * 0.1
    b2 = torch.randn(4, device="cuda", generator=g); bb = torch.randn(P, 4, device="cuda", generator=g)
    boxes = torch.zeros(rows, 4, device="cuda"); sig = torch.zeros(rows, 4, device="cuda")
    f = lambda: ops.box_final(h, w2, b2, bb, boxes, sig, rows, P, D)
    tw, tc = timed(f), timed(f, flush=flush)
    print(f"box_final rows {rows} D {D}: {tw:.1f} us warm / {tc:.1f} cold; checksums boxes {csum(boxes)} sig {csum(sig)}")
for n in (32 * 3 * 768 * 768, 16 * 3 * 840 * 840, 3 * 768 * 768, 1000003):
    src = torch.randn(n + 8, device="cuda", generator=g)[:n]; dst = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.cast_bf16(src, dst)
    tw, tc = timed(f), timed(f, flush=flush)
    print(f"cast n {n}: {tw:.1f} us warm / {tc:.1f} cold; checksum {csum(dst)}")
