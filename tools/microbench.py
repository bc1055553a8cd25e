"""Micro-benchmarks of the hot kernels at BASELINE shapes (B/16, batch 32). Prints TF/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops

DEV = "cuda"

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def bench_gemm(M, N, K, epi=ops.EPI_BIAS_BF16):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    bias = torch.randn(N, device=DEV)
    if epi == ops.EPI_RESID_F32:
        out = torch.zeros(ops.pad_rows(M), N, device=DEV); kw = dict(resid=out)
    else:
        out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16); kw = {}
    t = timeit(lambda: ops.gemm(epi, A, W, out, bias=bias, M=M, **kw))
    print(f"gemm M={M} N={N} K={K} epi={epi}: {t*1e3:.3f} ms  {2*M*N*K/t/1e12:.1f} TF/s", flush=True)

def bench_attn(B, H, T):
    Tp = (T + 7)//8*8; D = H*64; M = B*Tp
    qkv = torch.randn(ops.pad_rows(M), 3*D, device=DEV).bfloat16()
    vt = torch.randn(B*H*64*Tp + 128, device=DEV).bfloat16()
    out = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda: ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2*D:], 3*D, out, D, None, B, H, T, Tp, 0.125))   # the variant the model runs
    fl = 4.0*B*H*T*T*64
    print(f"attn B={B} H={H} T={T}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s", flush=True)

if __name__ == "__main__":
    """Warm the clocks first, then interleave the kernels (A/B/A/B): the first seconds of a fresh process run slower."""
    from owl_vit_object_detection_amd import _lib
    B = 32; Tp = 2312; M = B*Tp
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(30): bench_gemm(8192, 8192, 8192)
    shapes = [(M, 768, 768, ops.EPI_BIAS_BF16), (M, 1536, 768, ops.EPI_BIAS_BF16), (M, 3072, 768, ops.EPI_QGELU_BF16),
              (M, 768, 3072, ops.EPI_BIAS_BF16), (8192, 8192, 8192, ops.EPI_BIAS_BF16)]
    for sh in shapes:
        for rep in range(2):
            for tile in (256, 8):
                ops.GEMM_TILE = tile
                print("single-phase 256x256:" if tile == 256 else "ping-pong 256x256:  ", end=" ")
                bench_gemm(*sh)
    ops.GEMM_TILE = 0
    bench_attn(32, 12, 2305)
