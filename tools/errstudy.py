import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

from owl_vit_object_detection_amd import synth, weights, _lib
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.models import OwlViT
for cname, fx in (("owlvit-base-patch16", "f2_b16.npz"), ("owlvit-large-patch14", "f4_l14.npz")):
    cfg = get_config(cname)
    g = np.load(os.path.join(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"), fx))
    img = torch.from_numpy(synth.make_images(cfg, 1)).cuda()
    model = OwlViT(cfg, weights.make_weights(cfg), "cuda").eval()
    for dbg in (0, 1):
        _lib.call("owl_attention_debug", dbg)
        with torch.no_grad():
            pb, _, ps, _ = model(img)
        db = (pb[0].cpu() - torch.from_numpy(g["pred_boxes"][0])).abs()
        ds = (ps[0].cpu() - torch.from_numpy(g["pred_sims"][0])).abs()
        print(cname, "dbg", dbg, f"boxes max {db.max():.2e} rms {db.pow(2).mean().sqrt():.2e} | sims max {ds.max():.2e} rms {ds.pow(2).mean().sqrt():.2e}", flush=True)
    _lib.call("owl_attention_debug", 0)
