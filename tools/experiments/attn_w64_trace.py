"""s_memtime anatomy of one tile step of the one-wave-per-SIMD attention forward : workgroup 0, tiles 16..19 (variant 4 = stamped build, OWL_TUNING only), per wave:
cycles in [barrier + waits] / [LDS-DMA issue] / [phase A: 16 QK^T MFMAs || softmax || V reads] / [phase B: 16 PV MFMAs || softmax || K reads]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"
VAR = 4            # the stamped kernel: OWL_TUNING=1 build only (csrc/build.sh with OWL_TUNING=1, and OWL_TUNING=1 in the environment)
B, H, T = 32, 12, 2305
Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16)
for _ in range(3):
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, None, B, H, T, Tp, 0.125, variant=VAR)
torch.cuda.synchronize()
ws = ops.attention_redo_ws(B, H, T, DEV)
tr = ws[8192:8192 + 64].view(4, 4, 4).cpu()
print("| tile | wave | sync (barrier + waits) | DMA issue (4 pieces) | phase A | phase B | step |\n|---|---|---|---|---|---|---|")
for j in range(4):
    for w in range(4):
        a = tr[j, w].tolist()
        print(f"| {16 + j} | {w} | {a[0]} | {a[1]} | {a[2]} | {a[3]} | {sum(a)} |")
print("phase B of tile 17, wave 0: cycles since phase start at each MFMA slot:", ws[8192 + 64: 8192 + 80].tolist())
