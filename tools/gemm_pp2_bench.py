"""Same-process A/B of the ping-pong GEMM schedules: tile 8 = four quadrant phases per K-tile (gemm_pp.hip), tile 7 = two phases of 16 MFMAs
(gemm_pp2.hip); bit equality + interleaved timing on the model's shapes and at 8192^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def case(name, M, N, K, epi, rounds=5, iters=10):
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    b = torch.randn(N, device=DEV)
    f32 = epi in (ops.EPI_F32, ops.EPI_ACC_F32)
    aux = torch.randn(ops.pad_rows(M), N, device=DEV).bfloat16() if epi in (ops.EPI_DQGELU_BF16, ops.EPI_DGELU_BF16) else None
    _gemm = ops.gemm
    def gemm(epi, A, W, o, bias=None, M=None, tile=None):
        return _gemm(epi, A, W, o, bias=None if aux is not None else bias, aux=aux, M=M, tile=tile)
    class _O:                                      # (local shim: same call shape for every epilogue)
        pass
    ops_gemm = gemm
    outs = {}
    for t in (8, 7):
        o = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
        ops_gemm(epi, A, W, o, bias=b, M=M, tile=t); outs[t] = o
    torch.cuda.synchronize()
    eq = torch.equal(outs[7], outs[8])
    o = outs[8]
    times = {8: [], 7: []}
    for _ in range(2):
        for t in (8, 7):
            for _ in range(iters): ops_gemm(epi, A, W, o, bias=b, M=M, tile=t)
    for r in range(rounds):
        for t in (8, 7):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(iters): ops_gemm(epi, A, W, o, bias=b, M=M, tile=t)
            e1.record(); torch.cuda.synchronize(); times[t].append(e0.elapsed_time(e1) / iters)
    fl = 2.0 * M * N * K
    m8, m7 = sorted(times[8])[rounds // 2], sorted(times[7])[rounds // 2]
    print(f"{name:28s} M={M} N={N} K={K}: bits equal {eq};  4-phase {m8*1e3:7.1f} us {fl/m8/1e9:5.0f} TF/s | 2-phase {m7*1e3:7.1f} us {fl/m7/1e9:5.0f} TF/s  ({(m8/m7-1)*100:+.1f} %)", flush=True)


if __name__ == "__main__":
    M = 32 * 2312
    case("QKV", M, 2304, 768, ops.EPI_BIAS_BF16)
    case("out-proj", M, 768, 768, ops.EPI_BIAS_BF16)
    case("fc1 (quick-GELU)", M, 3072, 768, ops.EPI_QGELU_BF16)
    case("fc2", M, 768, 3072, ops.EPI_BIAS_BF16)
    case("half batch QKV", M // 2, 2304, 768, ops.EPI_BIAS_BF16)
    case("half batch fc1", M // 2, 3072, 768, ops.EPI_QGELU_BF16)
    case("half batch fc2", M // 2, 768, 3072, ops.EPI_BIAS_BF16)
    case("L/14 fc1", 16 * 3608, 4096, 1024, ops.EPI_QGELU_BF16)
    case("dX through quick-GELU'", M, 3072, 768, ops.EPI_DQGELU_BF16)
    case("box head dense (GELU)", 32 * 2304, 768, 768, ops.EPI_GELU_BF16)
    case("box head dX (GELU')", 32 * 2304, 768, 768, ops.EPI_DGELU_BF16)
    case("class head (f32 out)", 32 * 2304, 512, 768, ops.EPI_F32)
    case("8192^3", 8192, 8192, 8192, ops.EPI_BIAS_BF16, iters=5)
