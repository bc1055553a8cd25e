"""Stand-alone timing of the class head's tail kernels at the headline sizes (B/16 batch 32: 73 728 rows x 512; L/14 batch 16: 57 600 x 768; batch 1), forward and
backward, cold (a 600 MB buffer rewritten between launches: the operands come from HBM as in the step) and warm, with exact checksums of every output
for same-bits A/B across two builds of libowlhip.so (profiles/r06_class_sims.md)."""
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
for rows, Dt, C in ((73728, 512, 10), (57600, 768, 10), (2304, 512, 10)):
    g = torch.Generator(device="cuda").manual_seed(0)
    e = torch.randn(rows, Dt, device="cuda", generator=g) * 0.7
    Q = torch.randn(3 * C, Dt, device="cuda", generator=g)
    qhat = torch.zeros(32, Dt, device="cuda"); qn = torch.zeros(32, device="cuda")
    ops.query_normalize(Q, qhat, qn, 3 * C, Dt)
    sims = torch.zeros(rows, C, device="cuda"); am = torch.zeros(rows, C, dtype=torch.uint8, device="cuda"); inv = torch.zeros(rows, device="cuda")
    fwd = lambda: ops.class_sims(e, qhat, sims, am, inv, rows, Dt, C)
    tw, tc = timed(fwd), timed(fwd, flush=flush)
    dsims = torch.randn(rows, C, device="cuda", generator=g)
    de = torch.zeros(rows, Dt, device="cuda", dtype=torch.bfloat16); G = torch.zeros(rows, 32, device="cuda", dtype=torch.bfloat16); eb = torch.zeros_like(de)
    bwd = lambda: ops.class_sims_bwd(dsims, sims, am, inv, e, qhat, de, G, eb, rows, Dt, C)
    bw, bc = timed(bwd), timed(bwd, flush=flush)
    print(f"rows {rows} Dt {Dt}: class_sims fwd {tw:.1f} us warm / {tc:.1f} cold; bwd {bw:.1f} warm / {bc:.1f} cold; "
          f"checksums sims {csum(sims)} argmax {csum(am)} inv {csum(inv)} de {csum(de)} G {csum(G)} e_bf16 {csum(eb)}")
