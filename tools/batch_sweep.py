"""Robustness sweep: a train step at many batch sizes (sub-batch streams kick in at 8; odd sizes split unevenly), each image's outputs compared with its batch-1
forward (batch invariance, bitwise) and the loss against the mean of the batch-1 losses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT

for arch, sizes in (("small", (1, 2, 3, 5, 7, 8, 9, 11, 16, 17, 31, 33, 64)), ("owlvit-base-patch16", (1, 7, 8, 9, 17, 33))):
    cfg = get_config(arch)
    model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
    Bmax = max(sizes)
    imgs = torch.from_numpy(synth.make_images(cfg, Bmax)).cuda()
    labels, boxes = synth.make_targets(cfg, Bmax, max_boxes=8)
    crit = PushPullLoss(cfg.n_classes, synth.class_scales(cfg, labels))
    L = [torch.from_numpy(l).cuda() for l in labels]; Bx = [torch.from_numpy(b).cuda() for b in boxes]
    single = []
    for i in range(Bmax):
        model.flat_grad.zero_()
        pb, _, ps, _ = model(imgs[i:i + 1])
        l = crit(ps, L[i:i + 1], pb, Bx[i:i + 1]); tot = l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]; tot.backward()
        single.append((pb.detach().clone(), ps.detach().clone(), float(tot)))
    for B in sizes:
        model.flat_grad.zero_()
        pb, _, ps, _ = model(imgs[:B])
        l = crit(ps, L[:B], pb, Bx[:B]); tot = l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]; tot.backward()
        torch.cuda.synchronize()
        ok = all(torch.equal(pb[i], single[i][0][0]) and torch.equal(ps[i], single[i][1][0]) for i in range(B))
        mean1 = sum(s[2] for s in single[:B]) / B
        fin = bool(torch.isfinite(model.flat_grad).all())
        print(f"{arch} B={B:3d} chunks={model._encoder_chunks(B)}: batch-invariant bits {ok}; loss {float(tot):.6f} vs mean of batch-1 losses {mean1:.6f}; grads finite {fin}", flush=True)
        assert ok and fin and abs(float(tot) - mean1) < 1e-4 * max(1.0, abs(mean1))
print("sweep ok")
