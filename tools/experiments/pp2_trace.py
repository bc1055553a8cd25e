"""Timeline of one K-tile of the two-phase ping-pong GEMM (OWL_TUNING build): workgroup 0 stamps s_memtime at every boundary of K-tile 4 of its first
tile (gemm_pp2_kernel<BIAS, TRACE>); per wave, segment lengths in ticks:
  LOAD A issue (8 B + 8 A fragment reads + 4 A pieces) | lgkmcnt wait | barrier | 16 MFMAs issued | barrier | LOAD B issue (8 A reads + 4-5 B pieces) | counted vmcnt wait | barrier | 16 MFMAs | barrier
Usage: OWL_TUNING=1 python tools/pp2_trace.py [M N K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops, _lib

DEV = "cuda"
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (73984, 2304, 768)))
TILE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
KT = int(sys.argv[5]) if len(sys.argv) > 5 else 4
LINES = int(sys.argv[6]) if len(sys.argv) > 6 else 0          # 1: the quad-contiguous-store epilogue (tuning builds)
_lib.call("owl_gemm_pp2_lines", LINES)
A = torch.randn(ops.pad_rows(M), K, device=DEV).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
bias = torch.randn(N, device=DEV)
out = torch.zeros(ops.pad_rows(M), N, device=DEV, dtype=torch.bfloat16)
trace = torch.zeros(128 + 768, dtype=torch.int64, device=DEV)
for _ in range(5):
    ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
tiles = ((M + 255) // 256) * ((N + 255) // 256)
print(f"whole call (automatic kernel choice, 20 back to back): {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TF/s; {tiles} tiles of 256 x 256 = {tiles / 256:.2f} per workgroup")
_lib.call("owl_gemm_pp2_trace_tile", TILE)
_lib.call("owl_gemm_pp2_trace_ktile", KT)
_lib.call("owl_gemm_pp2_trace", trace)
ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M)
torch.cuda.synchronize()
_lib.call("owl_gemm_pp2_trace", None)
wg = trace[128:].cpu().numpy()
full = trace[:128].cpu().numpy().reshape(8, 16)
t = full[:, :11]
t0 = t.min()
names = ["LOAD A + wait", "bar", "MFMA A issue", "bar", "LOAD B + wait", "bar", "MFMA B issue", "bar"]
idx = [0, 2, 3, 4, 5, 7, 8, 9, 10]
print(f"M={M} N={N} K={K}, K-tile {KT} of tile {TILE} of workgroup 0; one row per wave (group = wave >> 2); start = ticks after the first stamp of the workgroup")
print("wave start | " + " | ".join(f"{n:>13s}" for n in names) + " | K-tile")
for w in range(8):
    d = t[w] - t0
    segs = [int(d[idx[i + 1]] - d[idx[i]]) for i in range(8)]
    print(f"w{w} {int(d[0]):6d} | " + " | ".join(f"{s:13d}" for s in segs) + f" | {int(d[10] - d[0])}")
print(f"mean ticks per K-tile per wave {float((t[:, 10] - t[:, 0]).mean()):.0f} (64 MFMAs per SIMD = 2048 shader cycles)")
if KT == 15:      # the stamps recorded the epilogue
    print("epilogue of the tile, per wave (ticks): bias values read | row block 0 | 1 | 2 | 3 converted + stores issued")
    for w in range(8):
        e = t[w]
        print(f"w{w}: {int(e[1] - e[0]):5d} | " + " | ".join(f"{int(e[2 + i] - e[1 + i]):5d}" for i in range(4)))
tt = full[:, 11:16]
print("whole tile, per wave: tile start -> K-loop start | K-loop (all K-tiles) | conversion + store issue | the epilogue barrier | tile")
for w in range(8):
    a = tt[w]
    print(f"w{w}: {int(a[1] - a[0]):6d} | {int(a[2] - a[1]):7d} | {int(a[4] - a[2]):6d} | {int(a[3] - a[4]):6d} | {int(a[3] - a[0]):7d} ticks")
import numpy as np
dur = wg[:256] / 2390.0
r0 = wg[256:512].astype(np.float64) / 100.0; r1 = wg[512:768].astype(np.float64) / 100.0          # us (100 MHz)
base = r0.min()
print(f"all 256 workgroups of the stamped launch (s_memrealtime, chip-wide 100 MHz): first instruction at {np.percentile(r0 - base, [0, 25, 50, 75, 100]).round(1)} us "
      f"(min / quartiles / max); last instruction at {np.percentile(r1 - base, [0, 25, 50, 75, 100]).round(1)} us; own duration {np.percentile(r1 - r0, [0, 50, 100]).round(1)} us (s_memtime / 2390: {np.percentile(dur, [0, 50, 100]).round(1)})")
