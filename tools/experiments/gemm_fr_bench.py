"""Round 4: the free-running GEMM (csrc/gemm_fr.hip, tile 5: two independent four-wave 128 x 256 workgroups per CU, BK = 32, three LDS stages) against what ships
(tile 0 = two-phase ping-pong 256 x 256 + half-height remainder) and the single-phase reference kernel (tile 256): bit equality (repeated: race screen) and
interleaved timing on the model's shapes.   python tools/gemm_fr_bench.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def case(name, M, N, K, epi, rounds=5, iters=10, tiles=(0, 5), with_aux_out=False):
    torch.manual_seed(7)
    A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    b = torch.randn(N, device=DEV)
    f32 = epi in (ops.EPI_F32, ops.EPI_ACC_F32)
    aux_in = torch.randn(ops.pad_rows(M), N, device=DEV).bfloat16() if epi in (ops.EPI_DQGELU_BF16, ops.EPI_DGELU_BF16) else None

    def run(tile, o, aux_o=None):
        return ops.gemm(epi, A, W, o, bias=None if aux_in is not None else b, aux=aux_in if aux_in is not None else aux_o, M=M, tile=tile)
    new = lambda: torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
    ref = new(); ref_aux = new() if with_aux_out else None
    run(256, ref, ref_aux); torch.cuda.synchronize()
    bad = 0
    for rep in range(6):
        o = new(); ao = new() if with_aux_out else None
        run(5, o, ao); torch.cuda.synchronize()
        if not torch.equal(o, ref) or (with_aux_out and not torch.equal(ao, ref_aux)):
            bad += 1
            if bad == 1:
                d = (o.float() - ref.float()).abs()
                rows = torch.nonzero(d.max(1).values > 0).flatten()
                cols = torch.nonzero(d.max(0).values > 0).flatten()
                print(f"   MISMATCH {name}: {int((d > 0).sum())} elements, rows {rows[:8].tolist()}..{int(rows[-1])} ({rows.numel()}), cols {cols[:8].tolist()}..{int(cols[-1])} ({cols.numel()}), max {float(d.max()):.3e}")
    o = new()
    times = {t: [] for t in tiles}
    for _ in range(2):
        for t in tiles:
            for _ in range(iters): run(t, o)
    for r in range(rounds):
        for t in tiles:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(iters): run(t, o)
            e1.record(); torch.cuda.synchronize(); times[t].append(e0.elapsed_time(e1) / iters)
    fl = 2.0 * M * N * K
    med = {t: sorted(times[t])[rounds // 2] for t in tiles}
    base = med[tiles[0]]
    txt = " | ".join(f"tile {t}: {med[t]*1e3:7.1f} us {fl/med[t]/1e9:5.0f} TF/s ({(med[t]/base-1)*100:+.1f} %)" for t in tiles)
    print(f"{name:26s} M={M} N={N} K={K}: tile 5 bits == tile 256: {'yes' if bad == 0 else f'NO ({bad}/6)'};  {txt}", flush=True)
    return bad


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    M = 32 * 2312
    bad = 0
    # small / ragged shapes first (guards, non-persistent launch, K = 96: the minimum of three K-steps)
    bad += case("ragged small", 777, 264, 128, ops.EPI_BIAS_BF16, rounds=1, iters=2)
    bad += case("one tile", 128, 256, 768, ops.EPI_QGELU_BF16, rounds=1, iters=2)
    bad += case("QKV", M, 2304, 768, ops.EPI_BIAS_BF16)
    bad += case("out-proj", M, 768, 768, ops.EPI_BIAS_BF16)
    bad += case("fc1 (quick-GELU)", M, 3072, 768, ops.EPI_QGELU_BF16)
    bad += case("fc2", M, 768, 3072, ops.EPI_BIAS_BF16)
    if not quick:
        bad += case("fc1 + saved pre-act", M, 3072, 768, ops.EPI_QGELU_BF16, with_aux_out=True)
        bad += case("half batch QKV", M // 2, 2304, 768, ops.EPI_BIAS_BF16)
        bad += case("half batch fc1", M // 2, 3072, 768, ops.EPI_QGELU_BF16)
        bad += case("half batch fc2", M // 2, 768, 3072, ops.EPI_BIAS_BF16)
        bad += case("L/14 QKV", 16 * 3608, 3072, 1024, ops.EPI_BIAS_BF16)
        bad += case("L/14 fc1", 16 * 3608, 4096, 1024, ops.EPI_QGELU_BF16)
        bad += case("dX through quick-GELU'", M, 3072, 768, ops.EPI_DQGELU_BF16)
        bad += case("dX (K = 2304)", M, 768, 2304, ops.EPI_BIAS_BF16)
        bad += case("box head dense (GELU)", 32 * 2304, 768, 768, ops.EPI_GELU_BF16)
        bad += case("box head dX (GELU')", 32 * 2304, 768, 768, ops.EPI_DGELU_BF16)
        bad += case("class head (f32 out)", 32 * 2304, 512, 768, ops.EPI_F32)
        bad += case("8192^3", 8192, 8192, 8192, ops.EPI_BIAS_BF16, iters=5)
    print("mismatching cases:", bad)
