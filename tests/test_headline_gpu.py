"""-m gpu: assertions behind the BASELINE.json headline workloads themselves (VERDICT r01 "configs untested").

  configs[1]  owlvit-base-patch16 bf16, batch 8, forward            -> all 8 images vs the CPU oracle
  configs[2]  owlvit-base-patch16 bf16, batch 32, full train step   -> test_full_step_at_baseline_batch[b16]
  configs[4]  owlvit-large-patch14 840x840, batch 16 per GPU        -> test_full_step_at_baseline_batch[l14]
  (configs[3] and the 8-GPU half of configs[4] are N independent replicas of these plus one all-reduce.)

The reference cannot run at batch > 1 (ref src/losses.py:23-24,100-106); batch semantics are "mean over images of the
reference's batch-1 step" (fixture F3).  The batch-1 step is pinned to the reference by F2 / F4
(test_model_gpu.py), so the full-batch step is held to the batch-1 step of the SAME HIP path:
  (i)   batch invariance: image i inside the batch gives the bits of image i alone (images 0, 13, 31 / 0, 7, 15);
  (ii)  losses(batch) = mean_i losses(image i), flat_grad(batch) = mean_i flat_grad(image i) (ref main.py:84-90);
  (iii) the discrete decisions (assignment, post-spreading labels) and the four per-image loss terms of >= 4 images
        against the CPU oracle fed the very same predictions, and those images' forward against the oracle's forward;
  (iv)  the fused AdamW update of the batch gradient against the oracle's AdamW (ref main.py:56-60,91).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import owl_oracle as O  # noqa: E402  (checker only)
from owl_vit_object_detection_amd import synth, weights  # noqa: E402
from owl_vit_object_detection_amd.config import get_config  # noqa: E402
from owl_vit_object_detection_amd.losses import PushPullLoss  # noqa: E402
from owl_vit_object_detection_amd.models import OwlViT  # noqa: E402
from owl_vit_object_detection_amd.optim import FusedAdamW  # noqa: E402

DEV = "cuda"
KEYS = ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")
# bf16 bar of the north star is 1e-2; measured on the final build (tools/errstudy.py): boxes 2.0e-3,
# sims 9.1e-4 at B/16, 1.8e-3 / 7.3e-4 at L/14 -> asserted at about twice the measured error
TOL_BOXES, TOL_SIMS = 4e-3, 2e-3


def _maxerr(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def _targets(cfg, B):
    labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
    return labels, boxes, synth.class_scales(cfg, labels)


def _train_step(model, crit, img, labels, boxes):
    """ref main.py:74-90 (zero_grad .. backward) through the drop-in call surface; returns detached results."""
    model.flat_grad.zero_()
    pb, n1, ps, n2 = model(img)
    assert n1 is None and n2 is None
    losses = crit(ps, [torch.from_numpy(l).to(DEV) for l in labels], pb, [torch.from_numpy(b).to(DEV) for b in boxes])
    (losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]).backward()
    return pb.detach().clone(), ps.detach().clone(), {k: float(v) for k, v in losses.items()}, model.flat_grad.clone()


@pytest.mark.parametrize("arch,B,probe", [("owlvit-base-patch16", 32, (0, 13, 31, 7)), ("owlvit-large-patch14", 16, (0, 7, 15, 3))],
                         ids=["b16-batch32", "l14-batch16"])
def test_full_step_at_baseline_batch(arch, B, probe):
    cfg = get_config(arch)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Wnp = weights.make_weights(cfg)
    model = OwlViT(cfg, Wnp, DEV)
    imgs_np = synth.make_images(cfg, B)
    imgs = torch.from_numpy(imgs_np).to(DEV)
    labels, boxes, scales = _targets(cfg, B)
    crit = PushPullLoss(cfg.n_classes, scales)

    pbB, psB, lossB, gradB = _train_step(model, crit, imgs, labels, boxes)
    lastB = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in crit.last.items()}
    assert pbB.shape == (B, cfg.patches, 4) and psB.shape == (B, cfg.patches, cfg.n_classes)
    assert bool(torch.isfinite(pbB).all()) and bool(torch.isfinite(psB).all()) and bool(torch.isfinite(gradB).all())

    # ---- (i) + (ii): every image alone, through the same path ----------------------------------------------------
    loss_sum = {k: 0.0 for k in KEYS}
    grad_sum = torch.zeros_like(gradB, dtype=torch.float64)
    worst_inv = 0.0
    for i in range(B):
        pb1, ps1, l1, g1 = _train_step(model, crit, imgs[i:i + 1], labels[i:i + 1], boxes[i:i + 1])
        for k in KEYS:
            loss_sum[k] += l1[k]
        grad_sum += g1.double()
        d = max(_maxerr(pb1[0], pbB[i]), _maxerr(ps1[0], psB[i]))
        worst_inv = max(worst_inv, d)
        if i in probe[:3]:
            assert torch.equal(pb1[0], pbB[i]) and torch.equal(ps1[0], psB[i]), f"image {i}: batch-{B} output differs from batch-1 by {d:.3e}"
        # per-image loss terms of the batch run = that image's batch-1 losses
        per = lastB["per_image"][i].cpu()
        for j, k in enumerate(KEYS):
            assert float(per[j]) == pytest.approx(l1[k], rel=1e-5, abs=1e-7), (i, k)
    print(f"{arch} B={B}: worst |batch - batch1| over all images {worst_inv:.3e}")
    assert worst_inv == 0.0                   # every image of the batch carries the very bits of its batch-1 forward
    for k in KEYS:
        assert lossB[k] == pytest.approx(loss_sum[k] / B, rel=2e-5, abs=1e-7), (k, lossB[k], loss_sum[k] / B)
    mean_grad = (grad_sum / B).float()
    floor = 1e-3 * max(float(mean_grad[o: o + model.p(n).numel()].norm()) for n, o in model.flat_offsets.items())
    worst = 0.0
    for n, o in model.flat_offsets.items():
        k = model.p(n).numel()
        a, r = gradB[o: o + k], mean_grad[o: o + k]
        rel = float((a - r).norm() / max(float(r.norm()), floor))
        worst = max(worst, rel)
        assert rel < 2e-5, (n, rel)           # f32 summation order only (the 1/B upstream scale is a power of two: exact in bf16); measured 9e-7
    print(f"{arch} B={B}: worst rel-L2 |flat_grad(B) - mean_i flat_grad(i)| = {worst:.3e}")

    # ---- (iii) decisions + loss terms of >= 4 images vs the CPU oracle on the same predictions; forward vs the oracle ----
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    sc = torch.from_numpy(scales)
    for i in probe:
        n = len(labels[i])
        det = {}
        lo = O.push_pull_loss_one(psB[i].cpu(), torch.from_numpy(labels[i]), pbB[i].cpu(), torch.from_numpy(boxes[i]), cfg.n_classes, sc, det)
        assert np.array_equal(lastB["pred_idx"][i, :n].cpu().numpy(), det["pred_idx"].numpy()), i
        assert np.array_equal(lastB["tgt_idx"][i, :n].cpu().numpy(), det["tgt_idx"].numpy()), i
        assert np.array_equal(lastB["target_classes"][i].cpu().numpy(), det["target_classes"].numpy()), i
        per = lastB["per_image"][i].cpu()
        for j, k in enumerate(KEYS):
            assert float(per[j]) == pytest.approx(float(lo[k]), rel=1e-4, abs=1e-6), (i, k)
        rb, rs = O.model_forward(cfg, w, torch.from_numpy(imgs_np[i:i + 1]))
        eb, es = _maxerr(pbB[i], rb[0]), _maxerr(psB[i], rs[0])
        print(f"{arch} B={B} image {i}: forward vs oracle max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
        assert eb < TOL_BOXES and es < TOL_SIMS, (i, eb, es)

    # ---- (iv) the optimizer step on the batch gradient (ref main.py:56-60: lr 3e-6, wd 0.1) -------------------------
    opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)
    model.flat_grad.copy_(gradB)
    before = model.flat_param.clone()
    opt.step()
    z = torch.zeros_like(before).cpu()
    p_ref, _, _ = O.adamw_step(before.cpu(), gradB.cpu(), z, z.clone(), 1, lr=3e-6, wd=0.1)
    np.testing.assert_allclose(model.flat_param.cpu().numpy(), p_ref.numpy(), rtol=2e-6, atol=1e-7)
    assert float((model.flat_param - before).abs().max()) > 0
    assert torch.equal(model.flat_bf16, model.flat_param.bfloat16())      # compute copy refreshed by the same kernel


def test_forward_config1_all_eight_images_match_oracle():
    """BASELINE configs[1]: owlvit-base-patch16 bf16, batch 8, 768x768, HIP ViT + heads forward only vs the CPU logits."""
    cfg = get_config("owlvit-base-patch16")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Wnp = weights.make_weights(cfg)
    model = OwlViT(cfg, Wnp, DEV).eval()
    imgs_np = synth.make_images(cfg, 8)
    with torch.no_grad():
        pb, n1, ps, n2 = model(torch.from_numpy(imgs_np).to(DEV))
    assert n1 is None and n2 is None and pb.shape == (8, 2304, 4) and ps.shape == (8, 2304, 10)
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    worst_b = worst_s = 0.0
    for i in range(8):
        rb, rs = O.model_forward(cfg, w, torch.from_numpy(imgs_np[i:i + 1]))
        eb, es = _maxerr(pb[i], rb[0]), _maxerr(ps[i], rs[0])
        worst_b, worst_s = max(worst_b, eb), max(worst_s, es)
        assert eb < TOL_BOXES and es < TOL_SIMS, (i, eb, es)
    print(f"configs[1] batch-8 forward vs oracle: worst max|d boxes|={worst_b:.3e} max|d sims|={worst_s:.3e}")
