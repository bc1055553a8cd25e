import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, _lib
DEV = "cuda"
torch.manual_seed(0)
M = 32 * 2312
bad = 0
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # 0: the library's choice (what the model runs: two-phase kernel + half-height remainder); 7: the two-phase kernel on the whole problem; 8: four-phase ping-pong (tuning build)
TUNING = _lib.is_tuning_build()
SHAPES = [(3072, 768, ops.EPI_QGELU_BF16), (2304, 768, ops.EPI_BIAS_BF16), (768, 3072, ops.EPI_BIAS_BF16), (768, 768, ops.EPI_BIAS_BF16), (1536, 768, ops.EPI_BIAS_BF16), (3072, 768, ops.EPI_DQGELU_BF16)]
if TUNING:
    SHAPES.append((768, 768, ops.EPI_TRANS_BF16))      # (the transposing epilogue exists in tuning builds only)
for (N, K, epi) in SHAPES:
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = None if epi == ops.EPI_DQGELU_BF16 else torch.randn(N, device=DEV)
    aux = torch.randn(ops.pad_rows(M), N, device=DEV).bfloat16() if epi == ops.EPI_DQGELU_BF16 else None      # (round 5: rolling pre-activation prefetch in the epilogue)
    big = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.uint8)
    def run(tile):
        ops.GEMM_TILE = tile
        if epi == ops.EPI_TRANS_BF16:
            out = torch.zeros(32 * N * 2312 + 256, device=DEV, dtype=torch.bfloat16)
            ops.gemm(epi, A, W, out, bias=bias, M=M, N=N, K=K, Tp=2312)
        else:
            out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
            ops.gemm(epi, A, W, out, bias=bias, aux=aux, M=M)
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
# round 5: the patch embedding's gather loader (L/14: 14-pixel rows, 4-byte-aligned 16-byte pieces) under the same concurrent traffic
from owl_vit_object_detection_amd import weights
from owl_vit_object_detection_amd.config import get_config
for arch, B in (("owlvit-large-patch14", 8), ("owlvit-base-patch16", 16)):
    cfg = get_config(arch)
    S, ps, D, Tp, P = cfg.image_size, cfg.patch_size, cfg.hidden, cfg.tokens_padded, cfg.patches
    img = torch.randn(B, 3, S, S, device=DEV).bfloat16()
    wk = weights.patch_weight_gather_layout((torch.randn(D, 3, ps, ps, device=DEV) * 0.05).bfloat16(), ps).contiguous()
    pos = torch.randn(cfg.tokens, D, device=DEV)
    scratch = None if ps & (ps - 1) == 0 else ops.zeros_rows(B * P, wk.shape[1], torch.bfloat16, DEV)
    big = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.uint8)
    def runp(tile):
        x = ops.zeros_rows(B * Tp, D, torch.float32, DEV)
        ops.patch_embed(img, wk, pos, x, B, S, ps, D, Tp, scratch=scratch, tile=tile)
        return x
    ref = runp(256)
    s2 = torch.cuda.Stream()
    for it in range(100):
        with torch.cuda.stream(s2):
            big.add_(1)
        if not torch.equal(runp(0), ref):
            bad += 1; print("MISMATCH patch embed", arch, it)
    torch.cuda.synchronize()
    print("patch embed", arch, "done", flush=True)
# round 6: the quick-GELU epilogue WITH the tile it saves for the backward (paired 16-byte stores behind a counted wait: vmcnt(36 / 37) instead of vmcnt(4) at each
# tile's first K-tile) -- output and saved tile against the single-phase kernel, B/16 and L/14 fc1 shapes
for (Mq, N, K) in ((32 * 2312, 3072, 768), (16 * 3608, 4096, 1024)):
    A = torch.randn(ops.pad_rows(Mq), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16(); bias = torch.randn(N, device=DEV)
    big = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.uint8)
    def runq(tile):
        ops.GEMM_TILE = tile
        out = torch.zeros(ops.pad_rows(Mq), N, device=DEV, dtype=torch.bfloat16); aux = torch.zeros(ops.pad_rows(Mq), N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(ops.EPI_QGELU_BF16, A, W, out, bias=bias, aux=aux, M=Mq)
        ops.GEMM_TILE = 0
        return out, aux
    ro, ra = runq(256)
    s2 = torch.cuda.Stream()
    for it in range(100):
        with torch.cuda.stream(s2):
            big.add_(1)
        go, ga = runq(TILE)
        if not (torch.equal(go, ro) and torch.equal(ga, ra)):
            bad += 1; print("MISMATCH qgelu + saved tile", Mq, N, K, it)
    torch.cuda.synchronize()
    print("qgelu + saved tile", Mq, N, K, "done", flush=True)
print("mismatches:", bad)
