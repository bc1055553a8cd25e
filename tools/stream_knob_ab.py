"""In-process A/B of one of OwlViT's stream-schedule switches (a boolean attribute of the model; default head_streams: box head / class head on two
streams, forward and backward): same bits (losses, predictions, gradient bucket), alternating timed runs of the full train step on one box.
Usage: stream_knob_ab.py [arch batch [attribute]].  Results in profiles/r02_encoder_streams.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW
arch, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("owlvit-base-patch16", 32)
KNOB = sys.argv[3] if len(sys.argv) > 3 else "head_streams"
cfg = get_config(arch)
model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)
img = torch.from_numpy(synth.make_images(cfg, B)).cuda()
labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
lab = [torch.from_numpy(l).cuda() for l in labels]; box = [torch.from_numpy(b).cuda() for b in boxes]
crit = PushPullLoss(cfg.n_classes, synth.class_scales(cfg, labels))


def step(update=True):
    opt.zero_grad()
    pb, _, ps, _ = model(img)
    losses = crit(ps, lab, pb, box)
    (losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]).backward()
    if update: opt.step()
    return pb, ps, losses


snap = {}
for flag in (False, True):                       # bits: one step from identical state, no update
    setattr(model, KNOB, flag)
    pb, ps, losses = step(update=False)
    torch.cuda.synchronize()
    snap[flag] = (pb.clone(), ps.clone(), {k: float(v) for k, v in losses.items()}, model.flat_grad.clone())
a, b = snap[False], snap[True]
print("bits equal:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), a[2] == b[2], torch.equal(a[3], b[3]), flush=True)
for _ in range(3):
    for flag in (False, True):
        setattr(model, KNOB, flag)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{arch} batch {B} {KNOB}={flag}: {dt * 1e3:.3f} ms/step  {B / dt:.1f} img/s", flush=True)
