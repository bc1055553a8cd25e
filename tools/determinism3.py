import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops
torch.manual_seed(0)
H, T = 12, 2305
Tp = (T + 7)//8*8; D = H*64
B = 8; M = B*Tp
qkv = torch.zeros(ops.pad_rows(M), 3*D, device="cuda", dtype=torch.bfloat16)
qkv[:M] = torch.randn(M, 3*D, device="cuda").bfloat16()
vt = torch.zeros(B*H*64*Tp + 128, device="cuda", dtype=torch.bfloat16)
vt[:B*H*64*Tp] = torch.randn(B*H*64*Tp, device="cuda").bfloat16()
from owl_vit_object_detection_amd import _lib
for flags in (0, 1, 2, 4, 7):
  _lib.call('owl_attention_debug', flags)
  ref = None; bad = 0
  for it in range(12):
    out = torch.zeros(ops.pad_rows(M), D, device="cuda", dtype=torch.bfloat16)
    ops.attention_fwd(qkv, qkv[:, D:], 3*D, vt, H*64*Tp, out, D, None, B, H, T, Tp, 0.125)
    torch.cuda.synchronize()
    if ref is None: ref = out.clone(); continue
    d = (out.float() - ref.float()).abs()
    if float(d.max()) > 0: bad += 1
  print("flags", flags, "runs differing from first:", bad, "of 11")
