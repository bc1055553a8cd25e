"""Is the small-batch forward launch-bound?  Eval forward at batch 1 / 2 / 8, eager against a hipGraph replay of the same launches (torch.cuda.graph; the model's
side streams join the capture through their events).  Result: no -- 1.947 vs 1.927 ms at batch 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.models import OwlViT
cfg = get_config("owlvit-base-patch16")
for B in (1, 2, 8):
    model = OwlViT(cfg, weights.make_weights(cfg), "cuda").eval()
    img = torch.from_numpy(synth.make_images(cfg, B)).cuda()
    def eager():
        with torch.no_grad():
            return model(img)
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    ref = eager(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eager()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = eager()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
        print(f"batch {B}: eager {te*1e3:.3f} ms, graph replay {tg*1e3:.3f} ms, same bits {torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])}", flush=True)
    except Exception as e:
        print(f"batch {B}: eager {te*1e3:.3f} ms, capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
