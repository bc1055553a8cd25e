"""Round 6, tuning build (tools/experiments/build.sh): dX through quick-GELU' (EPI 8 = acc x saved derivative) with the saved tile loaded and the result stored
through the quad-contiguous pattern (LINES: owl_gemm_pp2_lines(1)) against the accumulator-layout loads / stores (owl_gemm_pp2_lines(0)), one process, alternating;
bitwise against the single-phase reference kernel (tile 256).  A cold 512-MiB fill between launches optional (argv[1] == cold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from owl_vit_object_detection_amd import _lib, ops
assert _lib.is_tuning_build(), "needs the tuning build (tools/experiments/build.sh)"
cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
fill = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if cold else None
for name, M, N, K in (("B/16 dX through quick-GELU'", 32 * 2312, 3072, 768), ("B/16 half batch", 16 * 2312, 3072, 768), ("L/14 dX through quick-GELU'", 16 * 3608, 4096, 1024), ("ragged M = 1000, N = 520", 1000, 520, 256)):
    torch.manual_seed(1)
    A = torch.randn(ops.pad_rows(M), K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    aux = torch.randn(ops.pad_rows(M), N, device="cuda").bfloat16()
    o = torch.zeros(ops.pad_rows(M), N, device="cuda", dtype=torch.bfloat16); ref = torch.zeros_like(o)
    ops.gemm(ops.EPI_DQGELU_BF16, A, W, ref, aux=aux, M=M, tile=256)
    res, same = {0: [], 1: []}, {}
    for rnd in range(3):
        for lines in (0, 1):
            _lib.call("owl_gemm_pp2_lines", 3 if lines else 1)
            o.zero_(); ops.gemm(ops.EPI_DQGELU_BF16, A, W, o, aux=aux, M=M, tile=7); torch.cuda.synchronize()
            same[lines] = torch.equal(o[:M], ref[:M])
            ts = []
            for _ in range(7):
                if fill is not None: fill.fill_(1)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                n = 1 if cold else 10
                for _ in range(n): ops.gemm(ops.EPI_DQGELU_BF16, A, W, o, aux=aux, M=M, tile=7)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
            res[lines].append(float(np.median(ts)))
    _lib.call("owl_gemm_pp2_lines", 1)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print(f"{name}{' (cold)' if cold else ''}: accumulator-layout {a:7.1f} us, quad-contiguous loads + stores {b:7.1f} us ({100 * (b / a - 1):+.1f} %); bits == reference: {same}   rounds {res}", flush=True)
