"""A/B of the experimental four-wave GEMM (tile = 4) against the ping-pong kernel: bit-equality + time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import io, contextlib, torch
import microbench as mb
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
M = 32 * 2312
def check(M, N, K, epi):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    outs = []
    for tile in (8, 4):
        ops.GEMM_TILE = tile
        out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(epi, A, W, out, bias=bias, M=M)
        outs.append(out)
    torch.cuda.synchronize()
    ops.GEMM_TILE = 0
    print(f"M={M} N={N} K={K} epi={epi}: bitwise equal = {torch.equal(outs[0], outs[1])}, max diff {(outs[0].float()-outs[1].float()).abs().max().item():.3g}", flush=True)
for sh in [(M, 768, 768, 0), (M, 3072, 768, 1), (M, 768, 3072, 0), (1000, 512, 256, 0), (8192, 8192, 8192, 0)]:
    check(*sh)
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(20): mb.bench_gemm(8192, 8192, 8192)
shapes = [(M, 768, 768, ops.EPI_BIAS_BF16), (M, 1536, 768, ops.EPI_BIAS_BF16), (M, 3072, 768, ops.EPI_QGELU_BF16),
          (M, 768, 3072, ops.EPI_BIAS_BF16), (8192, 8192, 8192, ops.EPI_BIAS_BF16)]
for sh in shapes:
    for rep in range(2):
        for tile in (8, 4):
            ops.GEMM_TILE = tile
            print("ping-pong:" if tile == 8 else "four-wave:", end=" ")
            mb.bench_gemm(*sh)
ops.GEMM_TILE = 0
