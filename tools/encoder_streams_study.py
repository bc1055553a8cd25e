"""Experiment behind OwlViT(encoder_streams=2): N independent models, each with a share of the batch on its own stream, against one model with the
whole batch (eval forward).  Results in profiles/r02_encoder_streams.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.models import OwlViT
arch, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("owlvit-base-patch16", 32)
cfg = get_config(arch)
W = weights.make_weights(cfg)
img = torch.from_numpy(synth.make_images(cfg, B)).cuda()
def bench(name, f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize(); print(f"{arch} {name}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per {B} images", flush=True)
m = OwlViT(cfg, W, "cuda").eval()
def one():
    with torch.no_grad(): m(img)
bench("one stream", one)
for splits in ([B // 2, B - B // 2], [B // 3, B // 3, B - 2 * (B // 3)], [B // 4] * 4, [B * 5 // 8, B - B * 5 // 8]):
    models = [OwlViT(cfg, W, "cuda").eval() for _ in splits]
    streams = [torch.cuda.Stream() for _ in splits]
    parts = []; o = 0
    for s in splits: parts.append(img[o:o + s].contiguous()); o += s
    def multi():
        with torch.no_grad():
            for mm, st, im in zip(models, streams, parts):
                with torch.cuda.stream(st): mm(im)
    bench(f"streams x batches {splits}", multi)
    del models
bench("one stream", one)
