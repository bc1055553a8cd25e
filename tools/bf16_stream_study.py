"""Experiment: residual stream of the frozen prefix in bf16 (OwlViT(..., bf16_stream=True)) -- forward error against
the reference fixtures F2 / F4, and the train step rate, with and without.  Result in profiles/r02_bf16_stream.md."""
import time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.models import OwlViT
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for cname, fx, stream in [(c, f, s) for c, f in (("owlvit-base-patch16", "f2_b16.npz"), ("owlvit-large-patch14", "f4_l14.npz")) for s in (False, True)]:
    cfg = get_config(cname); g = np.load(os.path.join(G, fx))
    img = torch.from_numpy(synth.make_images(cfg, 1)).cuda()
    model = OwlViT(cfg, weights.make_weights(cfg), "cuda", bf16_stream=stream).eval()
    with torch.no_grad():
        pb, _, ps, _ = model(img)
        B = 32 if "base" in cname else 16
        big = torch.from_numpy(synth.make_images(cfg, B)).cuda()
        for _ in range(3):
            model(big)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            model(big)
        torch.cuda.synchronize(); fwd_ms = (time.perf_counter() - t0) * 100
    db = (pb[0].cpu() - torch.from_numpy(g["pred_boxes"][0])).abs(); ds = (ps[0].cpu() - torch.from_numpy(g["pred_sims"][0])).abs()
    print(f"stream={'bf16' if model._bf16_stream else 'f32 '} {cname}: boxes max {db.max():.2e} rms {db.pow(2).mean().sqrt():.2e} | sims max {ds.max():.2e} rms {ds.pow(2).mean().sqrt():.2e} | eval forward batch {B}: {fwd_ms:.2f} ms", flush=True)
    del model
