"""owl_hungarian alone at the headline sizes (32 images x 2304 predictions x 1..16 targets; L/14: 16 x 3600; crowd: 100 targets): time per launch + exact checksums of
the assignment for same-results A/B across two builds.  Costs: seeded normal (every column distinct) and a tie-heavy integer set."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import ops, _lib


def csum(t):
    return int(t.contiguous().view(-1).view(torch.uint8).to(torch.int64).mul(torch.arange(t.numel() * t.element_size(), device=t.device) % 251 + 1).sum())


def timed(f, n=30):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(0)
for B, P, Nmax, ties in ((32, 2304, 16, False), (32, 2304, 16, True), (16, 3600, 16, False), (2, 2304, 100, False), (1, 2304, 16, False)):
    cost = torch.randn(B, Nmax, P, device="cuda", generator=g)
    if ties: cost = torch.randint(0, 3, (B, Nmax, P), device="cuda", generator=g).float()
    counts = (torch.randint(1, Nmax + 1, (B,), device="cuda", generator=g)).int(); counts[0] = Nmax
    labels = torch.zeros(B, Nmax, dtype=torch.int64, device="cuda")
    pi = torch.zeros(B, Nmax, dtype=torch.int64, device="cuda"); ti = torch.zeros_like(pi); tc = torch.zeros(B, P, dtype=torch.int64, device="cuda")
    f = lambda: _lib.call("owl_hungarian", ops.stream(), cost, labels, counts, pi, ti, tc, B, P, Nmax, 99)
    t = timed(f)
    print(f"hungarian B={B} P={P} Nmax={Nmax} ties={ties}: {t:.1f} us; checksums pred {csum(pi)} tgt {csum(ti)} classes {csum(tc)}")
