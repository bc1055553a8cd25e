"""PushPullLoss -- host-side mirror of reference src/losses.py.

Same constructor and call signature (`PushPullLoss(n_classes, scales)(pred_sims, labels, pred_boxes,
boxes) -> {"loss_ce","loss_bg","loss_bbox","loss_giou"}`, ref src/losses.py:10,71-116).  The whole
criterion (matcher, box losses, sequential label spreading, focal-modulated BCE, and their gradients)
runs in HIP kernels with zero host syncs; batch > 1 means "mean over images of the reference's batch-1
loss" (the reference itself cannot run at batch > 1).
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .matcher import HungarianMatcher, PackedTargets, box_iou, generalized_box_iou  # noqa: F401  (re-exported like the reference)


class _PushPullFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_sims, pred_boxes, crit, tg):
        B, P, C = pred_sims.shape
        dev = pred_sims.device
        sims = pred_sims.detach().contiguous().float()
        boxes = pred_boxes.detach().contiguous().float()
        tc, pred_idx, tgt_idx, _ = crit.matcher.match_packed(sims, boxes, tg)
        s = ops.stream()
        _lib.call("owl_spread_labels", s, boxes, tc, B, P, crit.background_label, 0.85)
        need = pred_sims.requires_grad or pred_boxes.requires_grad
        per_image = torch.empty(B, 4, device=dev)
        losses = torch.empty(4, device=dev)
        dsims = torch.empty(B, P, C, device=dev) if need else None
        dl1 = torch.empty(B, P, 4, device=dev) if need else None
        dgiou = torch.empty(B, P, 4, device=dev) if need else None
        _lib.call("owl_push_pull_loss", s, sims, boxes, tc, crit.scales, tg.boxes, pred_idx, tgt_idx, tg.counts, per_image,
                  losses, dsims, dl1, dgiou, B, P, C, tg.Nmax, crit.background_label)
        ctx.saved = (tc, dsims, dl1, dgiou, B, P, C, crit.background_label)
        crit.last = dict(target_classes=tc, pred_idx=pred_idx, tgt_idx=tgt_idx, per_image=per_image, sizes=tg.sizes)
        # four scalar outputs (not one 4-vector indexed afterwards: each index would cost a zero-fill + a copy in its backward)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        tc, dsims, dl1, dgiou, B, P, C, bg = ctx.saved
        dev = tc.device
        g = torch.stack([g0, g1, g2, g3]).to(device=dev, dtype=torch.float32)
        out_sims = torch.empty(B, P, C, device=dev)
        out_boxes = torch.empty(B, P, 4, device=dev)
        _lib.call("owl_push_pull_loss_bwd", ops.stream(), g, tc, dsims, dl1, dgiou, out_sims, out_boxes, B, P, C, bg)
        return out_sims, out_boxes, None, None


class PushPullLoss(nn.Module):
    def __init__(self, n_classes, scales):
        super().__init__()
        self.matcher = HungarianMatcher(n_classes)
        self.scales = None if scales is None else torch.as_tensor(scales, dtype=torch.float32)
        self.background_label = n_classes
        self.last = None

    def pack(self, target_classes, target_boxes, device):
        if isinstance(target_classes, PackedTargets):
            return target_classes
        labels = list(target_classes) if not torch.is_tensor(target_classes) else list(target_classes.unbind(0))
        boxes = list(target_boxes) if not torch.is_tensor(target_boxes) else list(target_boxes.unbind(0))
        return PackedTargets(labels, boxes, device, self.background_label)

    def forward(self, predicted_classes, target_classes, predicted_boxes, target_boxes=None):
        dev = predicted_classes.device
        if self.scales is not None and self.scales.device != dev:
            self.scales = self.scales.to(dev)
        tg = self.pack(target_classes, target_boxes, dev)
        if len(tg.sizes) != predicted_classes.shape[0]:
            raise ValueError("batch size mismatch between predictions and targets")
        ce, bg, l1, giou = _PushPullFn.apply(predicted_classes, predicted_boxes, self, tg)
        return {"loss_ce": ce, "loss_bg": bg, "loss_bbox": l1, "loss_giou": giou}
