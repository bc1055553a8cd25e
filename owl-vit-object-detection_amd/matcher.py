"""Box ops + Hungarian matcher -- host-side mirror of reference src/matcher.py.

`box_iou`, `generalized_box_iou` and `HungarianMatcher` keep the reference's names, argument order
and return structure.  The arithmetic runs in HIP kernels (csrc/loss.hip):
cost matrix, the rectangular assignment solve (one workgroup per image, replacing the reference's
`.cpu()` + scipy round trip at src/matcher.py:132-137) and the target scatter -- no host sync.
"""
import torch
import torch.nn as nn

from . import _lib, ops


class _PinnedRing:
    """A few pinned host staging buffers reused round-robin.  The host enqueues a step long before the GPU runs it, so a
    staging buffer may only be rewritten once the H2D copy that last read it has executed: each slot carries the event
    recorded behind its copy and is waited for (normally long done) before reuse."""

    def __init__(self, slots: int = 4):
        self.slots = [dict(buf=None, event=None) for _ in range(slots)]
        self.next = 0

    def take(self, nbytes: int):
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot["event"] is not None:
            slot["event"].synchronize()
        if slot["buf"] is None or slot["buf"].numel() < nbytes:
            slot["buf"] = torch.empty(max(nbytes, 4096), dtype=torch.uint8, pin_memory=True)
        return slot


_ring = {}


class PackedTargets:
    """Padded device form of the DETR-style target list (ref src/matcher.py:94-104 `targets`; ref main.py:77-79 labels / boxes):
    labels [B,Nmax] i64, boxes [B,Nmax,4] f32, counts [B] i32 (+ the host-side sizes, known without a sync).

    Host tensors are packed on the host into ONE pinned staging buffer and cross PCIe in ONE async copy; device tensors are
    concatenated and padded by one kernel (owl_pack_targets) -- never B small copies.  `n_classes` (optional) validates the
    label range where the labels are still host tensors (the reference raises IndexError for such a label, src/matcher.py:118);
    for device-resident labels the kernels refuse to index with a bad label and poison loss_ce with NaN instead."""

    def __init__(self, labels, boxes, device, n_classes=None):
        labels, boxes = list(labels), list(boxes)
        sizes = [int(l.shape[0]) for l in labels]
        if any(int(b.shape[0]) != n for b, n in zip(boxes, sizes)):
            raise ValueError("labels / boxes length mismatch")
        if not sizes or min(sizes) < 1:
            raise ValueError("every image needs at least one target box (the reference drops empty images, src/dataset.py:33)")
        device = torch.device(device)
        B, Nmax = len(sizes), max(sizes)
        self.sizes, self.Nmax = sizes, Nmax
        nl, nb = B * Nmax * 8, B * Nmax * 16
        on_host = all(not t.is_cuda for t in labels) and all(not t.is_cuda for t in boxes)
        if on_host:
            if n_classes is not None:
                for l in labels:
                    if l.numel() and (int(l.min()) < 0 or int(l.max()) >= n_classes):
                        raise IndexError(f"target label outside [0, {n_classes})")
            if device.type != "cuda":
                self.labels = torch.zeros(B, Nmax, dtype=torch.int64)
                self.boxes = torch.zeros(B, Nmax, 4, dtype=torch.float32)
                for b, (l, bx) in enumerate(zip(labels, boxes)):
                    self.labels[b, : sizes[b]] = l.to(torch.int64)
                    self.boxes[b, : sizes[b]] = bx.to(torch.float32)
                self.counts = torch.tensor(sizes, dtype=torch.int32)
                return
            slot = _ring.setdefault(device, _PinnedRing()).take(nl + nb + 4 * B)
            host = slot["buf"]
            host[: nl + nb + 4 * B].zero_()
            hl = host[:nl].view(torch.int64).view(B, Nmax)
            hb = host[nl: nl + nb].view(torch.float32).view(B, Nmax, 4)
            hc = host[nl + nb: nl + nb + 4 * B].view(torch.int32)
            for b, (l, bx) in enumerate(zip(labels, boxes)):
                hl[b, : sizes[b]] = l
                hb[b, : sizes[b]] = bx
                hc[b] = sizes[b]
            dev = torch.empty(nl + nb + 4 * B, dtype=torch.uint8, device=device)
            # (copy and event on `device`'s current stream -- which need not be the current device's)
            with torch.cuda.device(device):
                dev.copy_(host[: nl + nb + 4 * B], non_blocking=True)
                slot["event"] = torch.cuda.Event()
                slot["event"].record()
            self.labels = dev[:nl].view(torch.int64).view(B, Nmax)
            self.boxes = dev[nl: nl + nb].view(torch.float32).view(B, Nmax, 4)
            self.counts = dev[nl + nb:].view(torch.int32)
            return
        # device-resident lists (ref main.py:78-79 moves them with .to(device) before the criterion): concat + one pad kernel
        lab_cat = torch.cat([l.to(device=device, dtype=torch.int64).reshape(-1) for l in labels])
        box_cat = torch.cat([b.to(device=device, dtype=torch.float32).reshape(-1, 4) for b in boxes]).contiguous()
        slot = _ring.setdefault(device, _PinnedRing()).take(4 * (B + 1))
        ho = slot["buf"][: 4 * (B + 1)].view(torch.int32)
        acc = 0
        for b, n in enumerate(sizes):
            ho[b] = acc
            acc += n
        ho[B] = acc
        offsets = torch.empty(B + 1, dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            offsets.copy_(ho, non_blocking=True)
            slot["event"] = torch.cuda.Event()
            slot["event"].record()
        self.labels = torch.empty(B, Nmax, dtype=torch.int64, device=device)
        self.boxes = torch.empty(B, Nmax, 4, dtype=torch.float32, device=device)
        self.counts = torch.empty(B, dtype=torch.int32, device=device)
        _lib.call("owl_pack_targets", ops.stream(), lab_cat, box_cat, offsets, self.labels, self.boxes, self.counts, B, Nmax)

    @staticmethod
    def from_lists(targets, device, n_classes=None):
        return PackedTargets([t["labels"] for t in targets], [t["boxes"] for t in targets], device, n_classes)


def _pairwise(boxes1, boxes2, want):
    """IoU / union / GIoU [N,M] through the matching-cost kernel's box code is overkill for the eval
    helpers; these two free functions are thin f32 device kernels only used off the hot path, so they
    call the C ABI's pairwise entry."""
    N, M = boxes1.shape[0], boxes2.shape[0]
    out = torch.empty(3, N, M, dtype=torch.float32, device=boxes1.device)
    _lib.call("owl_box_pairwise", ops.stream(), boxes1.contiguous().float(), boxes2.contiguous().float(), out, N, M)
    return out


def box_iou(boxes1, boxes2):
    """ref src/matcher.py:8-21: returns (iou [N,M], union [N,M])."""
    o = _pairwise(boxes1, boxes2, "iou")
    return o[0], o[1]


def generalized_box_iou(boxes1, boxes2):
    """ref src/matcher.py:25-44 (xyxy boxes; degenerate boxes are rejected like the reference's asserts)."""
    if not (bool((boxes1[:, 2:] >= boxes1[:, :2]).all()) and bool((boxes2[:, 2:] >= boxes2[:, :2]).all())):
        raise AssertionError("degenerate boxes")
    return _pairwise(boxes1, boxes2, "giou")[2]


class HungarianMatcher(nn.Module):
    """ref src/matcher.py:48-159."""

    def __init__(self, n_classes, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.n_classes = n_classes
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def match_packed(self, pred_logits, pred_boxes, tg: PackedTargets):
        """Device-only core: returns (target_classes [B,P] i64, pred_idx [B,Nmax] i64, tgt_idx [B,Nmax] i64, costT)."""
        B, P, C = pred_logits.shape
        dev = pred_logits.device
        if tg.Nmax > P:
            raise ValueError("more targets than predictions is not supported")
        sims = pred_logits.detach().contiguous().float()
        boxes = pred_boxes.detach().contiguous().float()
        costT = torch.empty(B, tg.Nmax, P, dtype=torch.float32, device=dev)
        pred_idx = torch.empty(B, tg.Nmax, dtype=torch.int64, device=dev)      # (the solver zero-fills the padding itself)
        tgt_idx = torch.empty(B, tg.Nmax, dtype=torch.int64, device=dev)
        tc = torch.empty(B, P, dtype=torch.int64, device=dev)
        s = ops.stream()
        _lib.call("owl_match_cost", s, sims, boxes, tg.labels, tg.boxes, tg.counts, costT, B, P, C, tg.Nmax,
                  float(self.cost_class), float(self.cost_bbox), float(self.cost_giou))
        _lib.call("owl_hungarian", s, costT, tg.labels, tg.counts, pred_idx, tgt_idx, tc, B, P, tg.Nmax, self.n_classes)
        return tc, pred_idx, tgt_idx, costT

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Same contract as the reference: returns (target_classes, indices, idx)."""
        tg = targets if isinstance(targets, PackedTargets) else PackedTargets.from_lists(targets, outputs["pred_logits"].device, self.n_classes)
        tc, pred_idx, tgt_idx, _ = self.match_packed(outputs["pred_logits"], outputs["pred_boxes"], tg)
        indices = [(pred_idx[b, :n], tgt_idx[b, :n]) for b, n in enumerate(tg.sizes)]
        return tc, indices, self._get_src_permutation_idx(indices)
