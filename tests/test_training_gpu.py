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


@pytest.mark.parametrize("profile", ["init", "trained_like"])
def test_four_step_trajectory_matches_the_oracle(profile):
    """The whole loop -- forward, matcher, loss, backward, fused AdamW -- for four consecutive steps on the `small` config against the CPU oracle doing
    the same four steps (oracle.train_step + oracle.adamw_step): the loss trajectory and the parameters after the last step.  `trained_like` runs it
    on trained-checkpoint statistics (massive channels, wide LayerNorm gains, peaked attention; weights.py)."""
    from oracle import owl_oracle as O           # checker only
    from owl_vit_object_detection_amd import synth, weights
    from owl_vit_object_detection_amd.config import get_config
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import OwlViT
    from owl_vit_object_detection_amd.optim import FusedAdamW
    cfg = get_config("small")
    Wnp = weights.make_weights(cfg, profile=profile)
    B, lr, wd = 2, 2e-4, 0.1
    img = synth.make_images(cfg, B, seed=11)
    labels, boxes = synth.make_targets(cfg, B, max_boxes=6, seed=11)
    scales = synth.class_scales(cfg, labels)
    model = OwlViT(cfg, Wnp, "cuda")
    crit = PushPullLoss(cfg.n_classes, scales)
    opt = FusedAdamW(model, lr=lr, weight_decay=wd)
    image = torch.from_numpy(img).cuda()
    lab = [torch.from_numpy(l).cuda() for l in labels]; box = [torch.from_numpy(b).cuda() for b in boxes]
    hip = []
    for _ in range(4):
        opt.zero_grad()
        pb, _, ps, _ = model(image)
        l = crit(ps, lab, pb, box)
        tot = l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]
        tot.backward(); opt.step()
        hip.append(float(tot.detach()))
    w = {k: torch.from_numpy(v.copy()) for k, v in Wnp.items()}
    names = O.trainable_names(w)
    m = {n: torch.zeros_like(w[n]) for n in names}; v = {n: torch.zeros_like(w[n]) for n in names}
    ref = []
    for step in range(1, 5):
        _, lo, grads = O.train_step(cfg, w, torch.from_numpy(img), [torch.from_numpy(x) for x in labels], [torch.from_numpy(x) for x in boxes], torch.from_numpy(scales))
        ref.append(float(lo["loss_ce"] + lo["loss_bg"] + lo["loss_bbox"] + lo["loss_giou"]))
        for n in names:
            w[n], m[n], v[n] = O.adamw_step(w[n], grads[n], m[n], v[n], step, lr=lr, wd=wd)
    print(f"{profile}: loss trajectory HIP {np.round(hip, 4)} oracle {np.round(ref, 4)}")
    assert ref[-1] < ref[0]                                            # the steps do something
    for a, b in zip(hip, ref):
        assert abs(a - b) <= 1e-2 * abs(b), (hip, ref)
    # parameters after four steps: the UPDATE (p4 - p0) against the oracle's, tensor by tensor (Adam normalises the gradient, so a step is ~lr per element
    # wherever the gradient has a sign: the comparison is of update directions)
    worst = 1.0
    for n in names:
        if n.endswith("k_proj.bias"):
            continue                      # softmax is invariant to a key bias: the true gradient is zero and Adam turns its round-off into +-lr steps
        d_hip = model.p(n).detach().cpu().double() - torch.from_numpy(Wnp[n]).double()
        d_ref = w[n].double() - torch.from_numpy(Wnp[n]).double()
        cos = float((d_hip * d_ref).sum() / (d_hip.norm() * d_ref.norm() + 1e-30))
        print(f"     {n:58s} |update| {float(d_ref.norm()):.3e} cos {cos:.4f}")
        worst = min(worst, cos)
    print(f"{profile}: worst cosine between the four-step parameter updates = {worst:.4f}")
    assert worst > 0.94, worst         # measured: init 0.9696 (box head: the L1 kink of a matched row), trained_like 0.9951; losses within 5.1e-3 / 2e-5
