"""LayerNorm backward at the headline shapes in the three forms the step uses -- (A) trainable layer's LN2: residual gradient in, dx f32 + bf16 out, parameter gradients and
the column sums of dx; (B) trainable LN1: parameter gradients only; (C) a frozen layer above the trainable one (L/14): dx only -- cold (600 MB rewritten between launches),
with exact checksums of every output for same-bits A/B across two builds of libowlhip.so."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import ops


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
g = torch.Generator(device="cuda").manual_seed(0)
for rows, D in ((32 * 2312, 768), (16 * 3608, 1024), (2312, 768), (777, 64)):
    x = torch.randn(rows, D, device="cuda", generator=g); dy = torch.randn(rows, D, device="cuda", generator=g).bfloat16()
    dres = torch.randn(rows, D, device="cuda", generator=g); gamma = torch.randn(D, device="cuda", generator=g)
    mu = x.mean(-1); rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt(); stats = torch.stack([mu, rstd], -1).contiguous()
    dx = torch.zeros(rows, D, device="cuda"); dxb = torch.zeros(rows, D, device="cuda", dtype=torch.bfloat16)
    out = []
    for name, kw in (("A full + params + colsum", dict(dres=dres, dx=dx, dxb=dxb, params=True, cs=True)), ("B params only", dict(dres=None, dx=None, dxb=None, params=True, cs=False)),
                     ("C dx only", dict(dres=dres, dx=dx, dxb=dxb, params=False, cs=False))):
        dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); cs = torch.zeros(D, device="cuda")
        f = lambda: ops.layernorm_bwd(dy, x, stats, gamma, kw["dres"], kw["dx"], dg if kw["params"] else None, db if kw["params"] else None, rows, D,
                                      dx_bf16=kw["dxb"], dx_colsum=cs if kw["cs"] else None)
        t = timed(f, flush=flush)
        dg.zero_(); db.zero_(); cs.zero_(); dx.zero_(); dxb.zero_(); f(); torch.cuda.synchronize()
        out.append(f"{name}: {t:.1f} us [{csum(dx)} {csum(dxb)} {csum(dg)} {csum(db)} {csum(cs)}]")
    print(f"rows {rows} D {D}: " + "; ".join(out))
