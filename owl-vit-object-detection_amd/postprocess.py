"""Inference post-process with the reference's call surface (ref src/models.py:122-146), on device.

``PostProcess(confidence_threshold, iou_threshold)(all_pred_boxes, pred_classes)`` -> ``(pred_boxes [1,K,4],
classes [1,K] i64, scores [1,K])`` exactly as the reference for its batch size of 1 (results ordered by descending
score, as ``torchvision.ops.batched_nms`` returns them).  Additions are keyword-only:

* ``top_k``  -- the eval loop's ``torch.topk(scores, min(200, K))`` (ref main.py:114-117) folded in: because the
  kept list is already score-descending it is a prefix, so the scan stops after ``top_k`` kept boxes;
* batch > 1 -- returns padded ``[B, Kmax, ...]`` tensors (classes padded with -1, scores with 0) and leaves the
  per-image counts in ``self.last_counts`` (device i32) -- with ``top_k`` given there is NO host sync at all.

* ``nms_route`` (constructor) -- ``torchvision.ops.batched_nms`` has two routes and picks by tensor size and device
  (torchvision/ops/boxes.py): per class on the raw coordinates, or all classes at once on ``boxes + class * (max + 1)``.
  The default ``"torchvision_gpu"`` does what torchvision does for a tensor on a GPU -- where the reference's eval loop
  runs (main.py:30,105-111): the coordinate trick up to 20 000 box coordinates past the threshold, i.e. always for
  OWL-ViT's <= 3600 boxes; ``"torchvision_cpu"`` mirrors the CPU path (per class above 1000 boxes); ``"per_class"`` /
  ``"coordinate_offset"`` pin one.  The routes only differ where the f32 rounding of the shifted coordinates moves an
  IoU across the threshold.

All arithmetic runs in ``owl_postprocess`` (csrc/postprocess.hip): one sort launch, one pair-mask launch, one scan
launch.  No CPU fallback.
"""
import torch

from . import ops


class PostProcess:
    def __init__(self, confidence_threshold=0.75, iou_threshold=0.3, *, nms_route="torchvision_gpu"):   # ref models.py:123-125 (same defaults)
        if nms_route not in ops.NMS_ROUTES:
            raise ValueError(f"PostProcess: nms_route must be one of {sorted(ops.NMS_ROUTES)}")
        self.confidence_threshold = confidence_threshold
        self.iou_threshold = iou_threshold
        self.nms_route = nms_route
        self.last_counts = None
        self.last_patch_idx = None

    def __call__(self, all_pred_boxes, pred_classes, *, top_k=None):
        if all_pred_boxes.dim() == 2:
            all_pred_boxes, pred_classes = all_pred_boxes[None], pred_classes[None]
        if all_pred_boxes.dim() != 3 or all_pred_boxes.shape[-1] != 4 or pred_classes.dim() != 3 \
                or pred_classes.shape[:2] != all_pred_boxes.shape[:2]:
            raise ValueError(f"PostProcess: expected boxes [B,P,4] and sims [B,P,C], got {tuple(all_pred_boxes.shape)} / {tuple(pred_classes.shape)}")
        B, P, C = pred_classes.shape
        max_out = int(top_k) if top_k is not None else P
        boxes, classes, scores, patch, counts = ops.postprocess(
            all_pred_boxes.detach().float().contiguous(), pred_classes.detach().float().contiguous(),
            max_out, float(self.confidence_threshold), float(self.iou_threshold), self.nms_route)
        self.last_counts, self.last_patch_idx = counts, patch
        if B == 1:                                  # the reference's contract: variable-length, one host read-back
            k = int(counts[0].item())
            self.last_patch_idx = patch[:, :k]
            return boxes[:, :k], classes[:, :k], scores[:, :k]
        if top_k is None:
            kmax = int(counts.max().item())
            self.last_patch_idx = patch[:, :kmax]
            return boxes[:, :kmax], classes[:, :kmax], scores[:, :kmax]
        return boxes, classes, scores
