"""-m gpu: the train loop of ref main.py:74-91 through the drop-in call surface actually learns (tiny config, fixed batch)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tiny_model_overfits_a_fixed_batch():
    from owl_vit_object_detection_amd import synth
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import load_model
    from owl_vit_object_detection_amd.optim import FusedAdamW
    model = load_model({str(i): i for i in range(4)}, "cuda", arch="tiny").train()
    cfg = model.cfg
    criterion = PushPullLoss(cfg.n_classes, None)
    optimizer = FusedAdamW(model, lr=3e-4, weight_decay=0.1)
    image = torch.from_numpy(synth.make_images(cfg, 4, seed=7)).cuda()
    labels, boxes = synth.make_targets(cfg, 4, max_boxes=8, seed=7)
    labels = [torch.from_numpy(l).cuda() for l in labels]
    boxes = [torch.from_numpy(b).cuda() for b in boxes]
    hist = []
    for it in range(80):
        optimizer.zero_grad()
        all_pred_boxes, pred_classes, pred_sims, _ = model(image)          # ref main.py:82
        losses = criterion(pred_sims, labels, all_pred_boxes, boxes)        # ref main.py:83
        loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
        loss.backward()
        optimizer.step()
        hist.append(float(loss.detach()))
    assert pred_classes is None and all(np.isfinite(hist))
    assert hist[-1] < 0.35 * hist[0], (hist[0], hist[-1])


def test_stock_adamw_drives_the_same_parameters():
    """ref main.py:56-60: torch.optim.AdamW(model.parameters(), ...) works unchanged on the flat-bucket views."""
    from owl_vit_object_detection_amd import synth
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import load_model
    model = load_model({str(i): i for i in range(4)}, "cuda", arch="tiny").train()
    cfg = model.cfg
    criterion = PushPullLoss(cfg.n_classes, None)
    optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-4, weight_decay=0.1)
    image = torch.from_numpy(synth.make_images(cfg, 2, seed=9)).cuda()
    labels, boxes = synth.make_targets(cfg, 2, max_boxes=6, seed=9)
    labels = [torch.from_numpy(l).cuda() for l in labels]
    boxes = [torch.from_numpy(b).cuda() for b in boxes]
    first = last = None
    for it in range(40):
        optimizer.zero_grad()
        pb, _, ps, _ = model(image)
        l = criterion(ps, labels, pb, boxes)
        loss = l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]
        loss.backward()
        optimizer.step()
        last = float(loss.detach())
        first = last if first is None else first
    assert np.isfinite(last) and last < 0.7 * first, (first, last)
