"""In-model A/B of per-call kernel choices that were made on stand-alone launches (inside the model the operands come out of the Infinity Cache in a
different state): attention forward plain vs class-token-peeled tiling, weight-gradient GEMM single-phase vs ping-pong, NT GEMM four-phase vs two-phase.
Same process, alternating timed runs of the full train step.  Usage: inmodel_ab.py [arch batch]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import ops, synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW
arch, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("owlvit-base-patch16", 32)
cfg = get_config(arch)
model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)
img = torch.from_numpy(synth.make_images(cfg, B)).cuda()
labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
lab = [torch.from_numpy(l).cuda() for l in labels]; box = [torch.from_numpy(b).cuda() for b in boxes]
crit = PushPullLoss(cfg.n_classes, synth.class_scales(cfg, labels))
_gemm, _tn = ops.gemm, ops.gemm_tn_slab
state = dict(tile=None, tn=0)
ops.gemm = lambda *a, **k: _gemm(*a, **{**k, "tile": k.get("tile") if k.get("tile") is not None else state["tile"]})
ops.gemm_tn_slab = lambda *a, **k: _tn(*a, **{**k, "variant": state["tn"]})


def step():
    opt.zero_grad()
    pb, _, ps, _ = model(img)
    losses = crit(ps, lab, pb, box)
    (losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]).backward()
    opt.step()


def timed(name):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{arch} batch {B} {name}: {dt * 1e3:.3f} ms/step  {B / dt:.1f} img/s", flush=True)


def setk(attn=0, tile=None, tn=0):
    ops.ATTN_VARIANT = attn; state["tile"] = tile; state["tn"] = tn


for _ in range(2):
    setk(); timed("shipped choices")
    setk(attn=1); timed("attention forward: plain tiling")
    setk(tn=1); timed("weight-gradient GEMM: single-phase")
    setk(tile=8); timed("NT GEMM: four-phase ping-pong")
setk(); timed("shipped choices")
