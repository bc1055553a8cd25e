"""fc1 with the quick-GELU epilogue at the headline shapes, without and with the tile it saves for the backward (aux = quick_gelu'(u); layers the backward passes
through only), + the box head's erf-GELU dense with its saved pre-activation: time per launch (cold: a 600 MB buffer rewritten between launches) and exact
checksums of both outputs for same-bits A/B across two builds of libowlhip.so (profiles/r06_fc1_aux.md)."""
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
for name, epi, M, N, K in (("fc1 B/16 batch 32", ops.EPI_QGELU_BF16, 32 * 2312, 3072, 768), ("fc1 L/14 batch 16", ops.EPI_QGELU_BF16, 16 * 3608, 4096, 1024),
                           ("fc1 B/16 batch 1 ", ops.EPI_QGELU_BF16, 2312, 3072, 768), ("box dense B/16    ", ops.EPI_GELU_BF16, 73728, 768, 768),
                           ("fc1 ragged        ", ops.EPI_QGELU_BF16, 777, 328, 256)):
    A = ops.zeros_rows(M, K, torch.bfloat16, "cuda"); A[:M] = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16(); bias = torch.randn(N, device="cuda", generator=g) * 0.1
    out = ops.zeros_rows(M, N, torch.bfloat16, "cuda"); aux = ops.zeros_rows(M, N, torch.bfloat16, "cuda")
    f0 = lambda: ops.gemm(epi, A, W, out, bias=bias, M=M)
    f1 = lambda: ops.gemm(epi, A, W, out, bias=bias, aux=aux, M=M)
    t0 = timed(f0, flush=flush); c0 = csum(out[:M]); out.zero_()
    t1 = timed(f1, flush=flush)
    print(f"{name} M={M} N={N} K={K}: {t0:.1f} us without aux, {t1:.1f} us with; checksums out {c0} / {csum(out[:M])} aux {csum(aux[:M])}")
