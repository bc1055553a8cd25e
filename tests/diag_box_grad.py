"""Diagnostic (GPU box, by hand): why do the end-to-end box-head gradients of the full-size fixtures differ from the reference's?
HIP step with retain_grad on (pred_boxes, pred_sims); the oracle's autograd of the loss AT THE HIP OUTPUTS; the fixture's decisions."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import owl_oracle as O
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT

arch, fx = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("owlvit-base-patch16", "f2_b16")
profile = sys.argv[3] if len(sys.argv) > 3 else "init"
cfg = get_config(arch); g = np.load(os.path.join(ROOT, "tests", "golden", fx + ".npz"))
img = synth.make_images(cfg, 1); labels, boxes = synth.make_targets(cfg, 1, max_boxes=16)
model = OwlViT(cfg, weights.make_weights(cfg, profile=profile), "cuda"); crit = PushPullLoss(cfg.n_classes, g["scales"])
pb, _, ps, _ = model(torch.from_numpy(img).cuda()); pb.retain_grad(); ps.retain_grad()
l = crit(ps, [torch.from_numpy(x).cuda() for x in labels], pb, [torch.from_numpy(x).cuda() for x in boxes])
(l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
n = len(labels[0])
pi, ti = crit.last["pred_idx"][0, :n].cpu().numpy(), crit.last["tgt_idx"][0, :n].cpu().numpy()
print("HIP assignment  (pred -> tgt):", list(zip(pi.tolist(), ti.tolist())))
print("fixture assignment           :", list(zip(g["pred_idx"].tolist(), g["tgt_idx"].tolist())))
sims = ps.detach().cpu()[0].clone().requires_grad_(True); bx = pb.detach().cpu()[0].clone().requires_grad_(True)
d = {}
lo = O.push_pull_loss_one(sims, torch.from_numpy(labels[0]), bx, torch.from_numpy(boxes[0]), cfg.n_classes, torch.from_numpy(g["scales"]), d)
(lo["loss_ce"] + lo["loss_bg"] + lo["loss_bbox"] + lo["loss_giou"]).backward()
print("oracle assignment at HIP outputs:", list(zip(d["pred_idx"].tolist(), d["tgt_idx"].tolist())))
gb, gs = pb.grad[0].cpu(), ps.grad[0].cpu()
print(f"d loss / d boxes at the HIP outputs: HIP vs oracle max abs diff {float((gb - bx.grad).abs().max()):.3e} (|oracle| max {float(bx.grad.abs().max()):.3e}); rows with gradient: HIP {int((gb.abs().sum(1) > 0).sum())} oracle {int((bx.grad.abs().sum(1) > 0).sum())}")
print(f"d loss / d sims  at the HIP outputs: HIP vs oracle max abs diff {float((gs - sims.grad).abs().max()):.3e} (|oracle| max {float(sims.grad.abs().max()):.3e})")
# the reference's own d_boxes at ITS outputs
rs = torch.from_numpy(g["pred_sims"][0]).clone().requires_grad_(True); rb = torch.from_numpy(g["pred_boxes"][0]).clone().requires_grad_(True)
lr = O.push_pull_loss_one(rs, torch.from_numpy(labels[0]), rb, torch.from_numpy(boxes[0]), cfg.n_classes, torch.from_numpy(g["scales"]))
(lr["loss_ce"] + lr["loss_bg"] + lr["loss_bbox"] + lr["loss_giou"]).backward()
a, b = gb.flatten().double(), rb.grad.flatten().double()
print(f"d_boxes: HIP (at its outputs) vs reference (at its outputs): cos {float((a * b).sum() / (a.norm() * b.norm())):.4f}, norms {float(a.norm()):.4f} / {float(b.norm()):.4f}")
rows = torch.nonzero(rb.grad.abs().sum(1) > 0).flatten()
for r in rows.tolist():
    t = int(g["tgt_idx"][list(g["pred_idx"]).index(r)]) if r in list(g["pred_idx"]) else -1
    print(f"  row {r:5d} tgt {t:2d}: ref d_box {rb.grad[r].numpy().round(4)} HIP d_box {gb[r].numpy().round(4)} | ref box - tgt {(rb[r].detach() - torch.from_numpy(boxes[0][t])).numpy().round(4) if t >= 0 else None} HIP box - tgt {(bx[r].detach() - torch.from_numpy(boxes[0][t])).numpy().round(4) if t >= 0 else None}")
