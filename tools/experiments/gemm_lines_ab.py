"""OWL_TUNING build: the two-phase ping-pong GEMM's forward bf16 epilogues with quad-contiguous stores (W rows permuted in LDS + in-register quad
transposition, csrc/gemm_common.h epi_lines_bf16) against the accumulator-layout stores; one process, alternating; outputs compared bitwise."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(73984, 2304, 768), (73984, 768, 768), (73984, 768, 3072), (73984, 3072, 768), (36992, 2304, 768), (36992, 3072, 768), (57616, 3072, 1024), (1000, 776, 256)]
for (M, N, K) in shapes:
    torch.manual_seed(M + N)
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    epi = ops.EPI_QGELU_BF16 if N == 3072 else ops.EPI_BIAS_BF16
    outs = []
    for on in (0, 2):
        _lib.call("owl_gemm_pp2_lines", on)
        out = torch.full((ops.pad_rows(M), N), 7.0, device=DEV, dtype=torch.bfloat16)
        aux = torch.zeros_like(out) if epi == ops.EPI_QGELU_BF16 else None
        ops.gemm(epi, A, W, out, bias=bias, aux=aux, M=M)
        outs.append((out, aux))
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    out = outs[1][0]
    res = []
    for rep in range(3):
        _lib.call("owl_gemm_pp2_lines", 0); a = t(lambda: ops.gemm(epi, A, W, out, bias=bias, M=M))
        _lib.call("owl_gemm_pp2_lines", 2); b = t(lambda: ops.gemm(epi, A, W, out, bias=bias, M=M))
        res.append(f"{a:.1f} -> {b:.1f} us ({(b / a - 1) * 100:+.1f} %)")
    print(f"M={M} N={N} K={K} {'qgelu' if epi == ops.EPI_QGELU_BF16 else 'bias'}: bitwise {'same' if same else 'DIFFERENT'}; " + "; ".join(res), flush=True)
_lib.call("owl_gemm_pp2_lines", 1)
