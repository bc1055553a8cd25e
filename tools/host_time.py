"""Host enqueue time per train step vs GPU time per step (is the Python launch loop the limit at small batch?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from owl_vit_object_detection_amd import weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW

dev = torch.device("cuda", 0)
cfg = get_config("owlvit-base-patch16")
model = OwlViT(cfg, weights.make_weights(cfg), dev)
opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)
for B in (32, 8, 1):
    batches = bench.synth_batches(cfg, B, dev, 0)
    crit = PushPullLoss(cfg.n_classes, None)

    def step(i):
        img, tg, _ = batches[i % 2]
        opt.zero_grad()
        pb, _, ps, _ = model(img)
        l = crit(ps, tg, pb)
        (l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
        opt.step()

    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {B}: host enqueue {(t1-t0)*100:.2f} ms/step, total {(t2-t0)*100:.2f} ms/step ({B*10/(t2-t0):.0f} img/s)", flush=True)
