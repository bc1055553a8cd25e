"""Time the device post-process at eval shapes; CPU oracle timed beside it on one image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from owl_vit_object_detection_amd.postprocess import PostProcess
from oracle import owl_oracle as O

def case(B, P, C, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((B, P, 2)).astype(np.float32) * 0.7
    wh = rng.random((B, P, 2)).astype(np.float32) * 0.3 + np.float32(0.02)
    return np.concatenate([c, c + wh], 2), (rng.random((B, P, C)).astype(np.float32) * 2 - 1) * np.float32(0.6)

for (B, P, top_k) in [(1, 2304, None), (32, 2304, 200), (32, 2304, None), (16, 3600, 200)]:
    boxes, sims = case(B, P, 10, 1)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(sims).cuda()
    pp = PostProcess(0.01, 0.6)
    for _ in range(3): pp(b, s, top_k=top_k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pp(b, s, top_k=top_k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    t0 = time.perf_counter(); O.post_process(boxes[0], sims[0], 0.01, 0.6, top_k=top_k); cpu = time.perf_counter() - t0
    alg = B * P * (10 * 4 + 16)
    print(f"B={B} P={P} top_k={top_k}: {ms:.3f} ms/call = {B/ms*1e3:.0f} img/s (kept {pp.last_counts.float().mean().item():.0f}/img); "
          f"input bytes {alg/1e6:.2f} MB; CPU oracle {cpu*1e3:.1f} ms/img", flush=True)
