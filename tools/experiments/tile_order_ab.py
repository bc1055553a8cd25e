"""OWL_TUNING build: train-step time by the tile order of the two-phase GEMM (column-block width of the fc1 / QKV launches), one process, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from owl_vit_object_detection_amd import _lib, weights, synth
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW
cfg = get_config("owlvit-base-patch16"); B = 32
model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
img = torch.from_numpy(synth.make_images(cfg, B)).cuda()
labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
lab = [torch.from_numpy(l).cuda() for l in labels]; box = [torch.from_numpy(b).cuda() for b in boxes]
crit = PushPullLoss(cfg.n_classes, synth.class_scales(cfg, labels)); opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)
def step():
    opt.zero_grad(); pb, _, ps, _ = model(img); l = crit(ps, lab, pb, box)
    (l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward(); opt.step()
def run(n=8):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for s in (2, 1):
    model.encoder_streams = s
    for epi, name, widths in ((1, "fc1 (quick-GELU, 12 column tiles)", (0, 12, 6, 4, 3, 2)), (0, "bias (QKV 9 / out-proj, fc2 3 column tiles)", (0, 9, 3, 1))):
        res = {}
        for rnd in range(3):
            for bw in widths:
                _lib.call("owl_gemm_pp2_block_width", epi, bw)
                res.setdefault(bw, []).append(run())
        _lib.call("owl_gemm_pp2_block_width", epi, 0)
        print(f"encoder_streams={s}, {name}: " + "; ".join(f"{'rule' if k == 0 else k}: {sorted(v)[1]:.2f} ms/step" for k, v in res.items()), flush=True)
