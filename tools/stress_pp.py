import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
torch.manual_seed(0)
M = 32 * 2312
bad = 0
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # 8: four-phase ping-pong; 7: the two-phase kernel (what the model runs); 0: the library's choice
for (N, K, epi) in [(3072, 768, ops.EPI_QGELU_BF16), (2304, 768, ops.EPI_BIAS_BF16), (768, 3072, ops.EPI_BIAS_BF16), (768, 768, ops.EPI_BIAS_BF16), (1536, 768, ops.EPI_BIAS_BF16), (768, 768, ops.EPI_TRANS_BF16)]:
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    big = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.uint8)
    def run(tile):
        ops.GEMM_TILE = tile
        if epi == ops.EPI_TRANS_BF16:
            out = torch.zeros(32 * N * 2312 + 256, device=DEV, dtype=torch.bfloat16)
            ops.gemm(epi, A, W, out, bias=bias, M=M, N=N, K=K, Tp=2312)
        else:
            out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
            ops.gemm(epi, A, W, out, bias=bias, M=M)
        ops.GEMM_TILE = 0
        return out
    ref = run(256)
    s2 = torch.cuda.Stream()
    for it in range(150):
        with torch.cuda.stream(s2):
            big.add_(1)                      # concurrent HBM traffic on another stream
        got = run(TILE)
        if not torch.equal(got, ref):
            bad += 1
            print("MISMATCH", N, K, epi, it, (got.float() - ref.float()).abs().max().item())
    torch.cuda.synchronize()
    print("shape", N, K, epi, "done", flush=True)
print("mismatches:", bad)
