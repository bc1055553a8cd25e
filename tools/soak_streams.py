"""Soak of the stream schedules at the headline size: N train steps of B/16 batch 32 (FusedAdamW, lr large enough to move the weights) with the default
schedule (two sub-batch streams, weight gradients on the side stream) and with one stream -- every loss of the trajectory and the final parameter bucket
must be bit-identical (any race between the streams shows up as a difference somewhere along 2 x N steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW

arch, B, steps = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("owlvit-base-patch16", 32, 40)
cfg = get_config(arch)
W = weights.make_weights(cfg)
imgs = [torch.from_numpy(synth.make_images(cfg, B, seed=s)).cuda() for s in (1, 2)]
tg = [synth.make_targets(cfg, B, max_boxes=16, seed=s) for s in (1, 2)]
scales = synth.class_scales(cfg, tg[0][0])


def run(streams):
    model = OwlViT(cfg, W, "cuda", encoder_streams=streams)
    crit = PushPullLoss(cfg.n_classes, scales)
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.1)
    traj = []
    for it in range(steps):
        labels, boxes = tg[it & 1]
        opt.zero_grad()
        pb, _, ps, _ = model(imgs[it & 1])
        l = crit(ps, [torch.from_numpy(x).cuda() for x in labels], pb, [torch.from_numpy(x).cuda() for x in boxes])
        (l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
        opt.step()
        traj.append(torch.stack([l[k].detach() for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")]))
    torch.cuda.synchronize()
    return torch.stack(traj).cpu(), model.flat_param.clone().cpu()


a, pa = run(2)
b, pb = run(1)
c, pc = run(2)
print(f"{arch} batch {B}, {steps} steps: first / last loss {a[0].sum():.4f} / {a[-1].sum():.4f}")
print("two streams vs one stream: losses equal", torch.equal(a, b), "parameters equal", torch.equal(pa, pb))
print("two streams, repeated:     losses equal", torch.equal(a, c), "parameters equal", torch.equal(pa, pc))
assert torch.equal(a, b) and torch.equal(pa, pb) and torch.equal(a, c) and torch.equal(pa, pc)
print("soak ok")
