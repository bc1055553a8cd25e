"""OWL_TUNING build, under rocprofv3 --pmc: a few launches of ONE GEMM of the train step with a given column-block width of the persistent tile walk.
usage: OWL_TUNING=1 python tools/gemm_counters.py <fc1|qkv|fc2|outproj|dqgelu> <block width, 0 = the launcher's rule> [launches]
Prints the mean HIP-event time per launch (meaningful only without the profiler attached)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import _lib, ops

shape, bw = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 32 * 2312
N, K, epi = {"fc1": (3072, 768, ops.EPI_QGELU_BF16), "qkv": (2304, 768, ops.EPI_BIAS_BF16), "fc2": (768, 3072, ops.EPI_BIAS_BF16),
             "outproj": (768, 768, ops.EPI_BIAS_BF16), "dqgelu": (3072, 768, ops.EPI_DQGELU_BF16)}[shape]
torch.manual_seed(1)
A = torch.randn(ops.pad_rows(M), K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = None if shape == "dqgelu" else torch.randn(N, device="cuda")
aux = torch.randn(ops.pad_rows(M), N, device="cuda").bfloat16() if shape == "dqgelu" else None
out = torch.zeros(ops.pad_rows(M), N, device="cuda", dtype=torch.bfloat16)
if epi in (ops.EPI_BIAS_BF16, ops.EPI_QGELU_BF16):
    _lib.call("owl_gemm_pp2_block_width", 1 if epi == ops.EPI_QGELU_BF16 else 0, bw)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")       # 512 MiB: past the Infinity Cache, so that every launch starts cold as in the step
def launch():
    flush.fill_(1)
    ops.gemm(epi, A, W, out, bias=bias, aux=aux, M=M, tile=7 if shape in ("fc1", "qkv", "dqgelu") else 0)
for _ in range(2): launch()
ts = []
for _ in range(n):
    flush.fill_(1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    ops.gemm(epi, A, W, out, bias=bias, aux=aux, M=M, tile=7 if shape in ("fc1", "qkv", "dqgelu") else 0)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print(f"{shape} bw={bw}: {sorted(ts)[len(ts) // 2]:.1f} us per launch (median of {n}, cold caches)", flush=True)
