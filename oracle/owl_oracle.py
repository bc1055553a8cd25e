"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32) restatement of the reference hot path: ``OwlViT.forward`` ->
``PushPullLoss.forward`` (-> ``HungarianMatcher.forward``) -> backward.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product
package never does (it must fail loudly when the HIP library is missing instead).

Self-contained: torch + numpy + the C solver in ``oracle/lsap.c`` -- no ``transformers``, no
``/root/reference`` import, so it travels to the GPU box.  Pinned against the reference itself by
``tests/golden/make_golden.py`` (run in the build container, where ``/root/reference`` and
``transformers`` exist): fixtures F1-F5 under ``tests/golden`` hold reference outputs for weights /
inputs this repo regenerates bit-identically from a seed; ``tests/test_oracle_golden.py`` checks this
file against them.  Version note: the reference pins transformers==4.30.2; the container has 5.15.0
(eager attention forced) -- mathematically identical arithmetic for this path.

Citations: ``ref:`` = path under /root/reference ; ``HF5:`` = transformers 5.15.0
``models/owlvit/modeling_owlvit.py``.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LSAP = None


# ------------------------------------------------------------------------------------------------
# Hungarian (scipy `_lsap` restated in C, oracle/lsap.c)
# ------------------------------------------------------------------------------------------------
def build_lsap(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle_lsap.so")
    src = os.path.join(_HERE, "lsap.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return so


def _lsap_lib():
    global _LSAP
    if _LSAP is None:
        lib = ctypes.CDLL(build_lsap())
        lib.oracle_lsap.restype = ctypes.c_int
        lib.oracle_lsap.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _LSAP = lib
    return _LSAP


def linear_sum_assignment(cost) -> "tuple[np.ndarray, np.ndarray]":
    """ref: src/matcher.py:136 (scipy call).  cost: [nr, nc] array-like; solved in float64."""
    c = np.ascontiguousarray(np.asarray(cost, dtype=np.float64))
    nr, nc = c.shape
    k = min(nr, nc)
    a = np.empty(k, dtype=np.int64)
    b = np.empty(k, dtype=np.int64)
    rc = _lsap_lib().oracle_lsap(nr, nc, c.ctypes.data, a.ctypes.data, b.ctypes.data)
    if rc != 0:
        raise ValueError("cost matrix is infeasible" if rc == -1 else "matrix contains invalid numeric entries")
    return a, b


# ------------------------------------------------------------------------------------------------
# Model forward (ref: src/models.py:98-119 and the HF modules it wraps)
# ------------------------------------------------------------------------------------------------
def quick_gelu(x):
    """transformers activations.py:122-123."""
    return x * torch.sigmoid(1.702 * x)


def box_bias(grid: int) -> torch.Tensor:
    """HF5:1071-1104 `compute_box_bias` for a square grid; [P,4] f32, p = y*W + x."""
    coords = torch.arange(1, grid + 1, dtype=torch.float32)
    xx, yy = torch.meshgrid(coords, coords, indexing="xy")
    bc = torch.stack((xx, yy), dim=-1)
    bc[..., 0] /= grid
    bc[..., 1] /= grid
    bc = bc.view(-1, 2)
    bc = torch.clip(bc, 0.0, 1.0)
    coord_bias = torch.log(bc + 1e-4) - torch.log1p(-bc + 1e-4)
    size = torch.full_like(coord_bias, 1.0)
    size[..., 0] /= grid
    size[..., 1] /= grid
    size_bias = torch.log(size + 1e-4) - torch.log1p(-size + 1e-4)
    return torch.cat([coord_bias, size_bias], dim=-1)


def encoder_layer(x, w, pre, heads, eps, taps=None):
    """HF5:488-509 (layer), HF5:428-459 + HF5:377-402 (eager attention), HF5:463-475 (MLP)."""
    B, T, D = x.shape
    dh = D // heads
    h = F.layer_norm(x, (D,), w[pre + "layer_norm1.weight"], w[pre + "layer_norm1.bias"], eps)
    q = F.linear(h, w[pre + "self_attn.q_proj.weight"], w[pre + "self_attn.q_proj.bias"]).view(B, T, heads, dh).transpose(1, 2)
    k = F.linear(h, w[pre + "self_attn.k_proj.weight"], w[pre + "self_attn.k_proj.bias"]).view(B, T, heads, dh).transpose(1, 2)
    v = F.linear(h, w[pre + "self_attn.v_proj.weight"], w[pre + "self_attn.v_proj.bias"]).view(B, T, heads, dh).transpose(1, 2)
    att = torch.matmul(q, k.transpose(2, 3)) * (dh ** -0.5)
    att = torch.softmax(att, dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, T, D)
    if taps is not None:
        taps[pre + "ln1"] = h
        taps[pre + "attn"] = o
    x = x + F.linear(o, w[pre + "self_attn.out_proj.weight"], w[pre + "self_attn.out_proj.bias"])
    h2 = F.layer_norm(x, (D,), w[pre + "layer_norm2.weight"], w[pre + "layer_norm2.bias"], eps)
    g = quick_gelu(F.linear(h2, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"]))
    x = x + F.linear(g, w[pre + "mlp.fc2.weight"], w[pre + "mlp.fc2.bias"])
    if taps is not None:
        taps[pre + "out"] = x
    return x


def model_forward(cfg, w, image, taps=None):
    """ref: src/models.py:98-119.  ``w``: name -> f32 tensor (reference parameter names).
    Returns (pred_boxes [B,P,4] xyxy, pred_sims [B,P,C])."""
    D, eps, g = cfg.hidden, cfg.ln_eps, cfg.grid
    B = image.shape[0]
    # embeddings: HF5:334-344 (conv k=s=patch, no bias; class token first; learned positions)
    pe = F.conv2d(image, w["backbone.embeddings.patch_embedding.weight"], stride=cfg.patch_size)
    pe = pe.flatten(2).transpose(1, 2)
    cls = w["backbone.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + w["backbone.embeddings.position_embedding.weight"].unsqueeze(0)
    if taps is not None:
        taps["embed"] = x
    x = F.layer_norm(x, (D,), w["backbone.pre_layernorm.weight"], w["backbone.pre_layernorm.bias"], eps)  # HF5:742
    if taps is not None:
        taps["pre_ln"] = x
    for i in range(cfg.layers):
        x = encoder_layer(x, w, f"backbone.encoder.layers.{i}.", cfg.heads, eps, taps)
    # ref: src/models.py:80-86 -- post_layernorm on ALL tokens, class-token merge, second LN
    x = F.layer_norm(x, (D,), w["backbone.post_layernorm.weight"], w["backbone.post_layernorm.bias"], eps)
    x = x[:, 1:, :] * x[:, :1, :]
    feats = F.layer_norm(x, (D,), w["post_post_layernorm.weight"], w["post_post_layernorm.bias"], eps)
    if taps is not None:
        taps["feats"] = feats
    # box head: HF5:983-999 (erf GELU) + bias + sigmoid + center_to_corners (ref: src/models.py:65-73)
    b = F.gelu(F.linear(feats, w["box_head.dense0.weight"], w["box_head.dense0.bias"]))
    b = F.gelu(F.linear(b, w["box_head.dense1.weight"], w["box_head.dense1.bias"]))
    b = F.linear(b, w["box_head.dense2.weight"], w["box_head.dense2.bias"])
    b = torch.sigmoid(b + box_bias(g).to(b.dtype))
    cx, cy, bw, bh = b.unbind(-1)
    pred_boxes = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], dim=-1)
    # class head: ref: src/models.py:24-38 (eps placement reproduced literally)
    e = F.linear(feats, w["class_predictor.dense0.weight"], w["class_predictor.dense0.bias"])
    e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
    q = w["queries"] / torch.linalg.norm(w["queries"], dim=-1, keepdim=True) + 1e-6
    sims = e @ q.transpose(1, 2)
    pred_sims = F.max_pool1d(sims, kernel_size=3, stride=3)
    return pred_boxes, pred_sims


# ------------------------------------------------------------------------------------------------
# Box ops / matcher / loss (ref: src/matcher.py, src/losses.py)
# ------------------------------------------------------------------------------------------------
def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(boxes1, boxes2):
    """ref: src/matcher.py:8-21."""
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union, union


def generalized_box_iou(boxes1, boxes2):
    """ref: src/matcher.py:25-44."""
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


@torch.no_grad()
def match_one(sims, boxes, labels, tgt_boxes, n_classes):
    """ref: src/matcher.py:85-159 for ONE image (the reference is batch-1).
    sims [P,C], boxes [P,4], labels [n] i64, tgt_boxes [n,4].
    Returns (cost [P,n] f32, pred_idx [n] ascending, tgt_idx [n], target_classes [P] i64)."""
    out_prob = sims.softmax(-1)
    cost_class = -out_prob[:, labels]
    cost_bbox = torch.cdist(boxes, tgt_boxes, p=1)
    cost_giou = -generalized_box_iou(boxes, tgt_boxes)
    C = 1 * cost_bbox + 1 * cost_class + 1 * cost_giou
    i, j = linear_sum_assignment(C.cpu().numpy())
    i = torch.as_tensor(i, dtype=torch.int64)
    j = torch.as_tensor(j, dtype=torch.int64)
    tc = torch.full((sims.shape[0],), n_classes, dtype=torch.int64)
    tc[i] = labels[j]
    return C, i, j, tc


@torch.no_grad()
def spread_labels(pred_boxes, target_classes, bg, thr=0.85):
    """ref: src/losses.py:100-106 -- sequential, in-place, transitive."""
    tc = target_classes.clone()
    P = tc.shape[0]
    for p in range(P):
        lab = int(tc[p])
        if lab == bg:
            continue
        iou, _ = box_iou(pred_boxes[p:p + 1], pred_boxes)
        tc[iou[0] > thr] = lab
    return tc


def class_loss(sims, tc, bg, scales):
    """ref: src/losses.py:16-40 for one image.  sims [P,C], tc [P]."""
    a = torch.abs(sims)
    pos = tc != bg
    pred_logits, bg_logits = a[pos], a[~pos]
    pos_t = F.one_hot(tc[pos], bg).float()
    neg_t = torch.zeros_like(bg_logits)
    pos_l = F.binary_cross_entropy(pred_logits, pos_t, weight=scales, reduction="none")
    neg_l = F.binary_cross_entropy(bg_logits, neg_t, weight=scales, reduction="none")
    pos_l = (torch.pow(1 - torch.exp(-pos_l), 2) * pos_l).sum(dim=1).mean()
    neg_l = (torch.pow(1 - torch.exp(-neg_l), 2) * neg_l).sum(dim=1).mean()
    return pos_l, neg_l


def push_pull_loss_one(sims, labels, boxes, tgt_boxes, n_classes, scales=None, detail=None):
    """ref: src/losses.py:71-116 at its native batch size of one."""
    C, i, j, tc = match_one(sims.detach(), boxes.detach(), labels, tgt_boxes, n_classes)
    num_boxes = labels.shape[0]
    src = boxes[i]
    tgt = tgt_boxes[j]
    loss_bbox = F.l1_loss(src, tgt, reduction="none").sum() / num_boxes
    loss_giou = (1 - torch.diag(generalized_box_iou(src, tgt))).sum() / num_boxes
    tc2 = spread_labels(boxes.detach(), tc, n_classes)
    loss_ce, loss_bg = class_loss(sims, tc2, n_classes, scales)
    if detail is not None:
        detail.update(cost=C, pred_idx=i, tgt_idx=j, target_classes_matched=tc, target_classes=tc2)
    return {"loss_ce": loss_ce, "loss_bg": loss_bg, "loss_bbox": loss_bbox, "loss_giou": loss_giou}


def push_pull_loss(pred_sims, labels, pred_boxes, tgt_boxes, n_classes, scales=None, details=None):
    """Batched semantics of this build: mean over images of the reference's
    batch-1 loss, term by term.  ``labels`` / ``tgt_boxes``: per-image lists."""
    B = pred_sims.shape[0]
    acc = None
    for b in range(B):
        d = {} if details is not None else None
        l = push_pull_loss_one(pred_sims[b], labels[b], pred_boxes[b], tgt_boxes[b], n_classes, scales, d)
        if details is not None:
            details.append(d)
        acc = l if acc is None else {k: acc[k] + l[k] for k in acc}
    return {k: v / B for k, v in acc.items()}


# ------------------------------------------------------------------------------------------------
# Train step (ref: main.py:74-91) and AdamW (ref: main.py:56-60)
# ------------------------------------------------------------------------------------------------
def trainable_names(w):
    keep = ("layers.11", "box", "post_layernorm", "class_predictor", "queries")   # ref: src/models.py:173-184
    return [n for n in w if any(s in n for s in keep)]


def train_step(cfg, w, image, labels, tgt_boxes, scales=None, with_grads=True):
    """One forward + loss + backward; returns (outputs, losses, grads{name: tensor})."""
    names = trainable_names(w)
    ww = {n: (t.detach().clone().requires_grad_(True) if n in names else t.detach()) for n, t in w.items()}
    boxes, sims = model_forward(cfg, ww, image)
    losses = push_pull_loss(sims, labels, boxes, tgt_boxes, cfg.n_classes, scales)
    total = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
    grads = {}
    if with_grads:
        total.backward()
        grads = {n: ww[n].grad for n in names}
    return (boxes.detach(), sims.detach()), {k: v.detach() for k, v in losses.items()}, grads


def adamw_step(p, g, m, v, step, lr=3e-6, wd=0.1, b1=0.9, b2=0.999, eps=1e-8):
    """Decoupled AdamW exactly as torch.optim.AdamW (single param group; ref: main.py:56-60)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# ---------------------------------------------------------------------------------------------------------------------
# Inference post-process: ref src/models.py:122-146 + the eval loop's top-200
# (ref main.py:114-117).  The suppression itself is ``torchvision.ops.batched_nms`` -- a third-party dependency that is
# neither under /root/reference nor installed here (unpinned in the reference's requirements.txt), so its PUBLISHED
# algorithm (torchvision/ops/boxes.py) is restated, BOTH of its routes:
#   * "per_class" (``_batched_nms_vanilla``): for each class, plain NMS on the raw coordinates of that class's boxes;
#   * "coordinate_offset" (``_batched_nms_coordinate_trick``): ``boxes + class * (boxes.max() + 1)`` in the boxes' dtype,
#     then ONE class-agnostic NMS over all boxes -- the same decisions except where the f32 rounding of the shifted
#     coordinates moves an IoU across the threshold;
#   * torchvision picks by size and device: per_class if ``boxes.numel() > 4000`` on CPU / ``> 20000`` on a GPU, else the
#     coordinate trick ("torchvision_cpu" / "torchvision_gpu" below).  The reference runs its eval loop on the GPU when
#     there is one (main.py:30,105-111): there every OWL-ViT output (<= 3600 boxes) takes the coordinate trick; on the
#     CPU path that BASELINE.json's parity is stated against, > 1000 boxes past the threshold take the per-class route.
# Plain NMS (both routes): visit boxes by descending score, keep a box iff no kept box has IoU > thr with it
# (IoU = inter / (area_a + area_b - inter), plain f32, intersection sides clamped at 0); result ordered by descending
# score.  Ties in score: lower patch index first (what a stable sort gives).
# PARITY NOTE: the max/threshold/index/shape part is pinned by running the reference's PostProcess (fixture F6, once per
# route); the NMS inside those runs is this same restatement injected as the torchvision stub => "parity unpinned" for
# torchvision's kernels themselves.
# ---------------------------------------------------------------------------------------------------------------------
def _nms_plain(boxes: np.ndarray, scores: np.ndarray, group, iou_threshold: float) -> np.ndarray:
    """Greedy NMS in f32 like torchvision's kernels; `group` (int array or None): only boxes of the same group suppress
    each other.  Indices kept, ordered by descending score (ties: ascending index)."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    n = boxes.shape[0]
    order = np.lexsort((np.arange(n), -scores.astype(np.float64)))     # primary: score desc; secondary: index asc
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    thr = np.float32(iou_threshold)
    suppressed = np.zeros(n, bool)
    keep = []
    for a in range(n):
        i = order[a]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        if rest.size == 0:
            break
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        hit = ovr > thr
        if group is not None:
            hit &= group[rest] == group[i]
        suppressed[rest[hit]] = True
    return np.asarray(keep, np.int64)


def nms_class_aware(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision ``_batched_nms_vanilla``: per class, raw coordinates."""
    return _nms_plain(boxes, scores, np.asarray(classes), iou_threshold)


def nms_coordinate_offset(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision ``_batched_nms_coordinate_trick``: offsets = class * (max coordinate + 1) (f32), one class-agnostic NMS."""
    boxes = np.asarray(boxes, np.float32)
    if boxes.shape[0] == 0:
        return np.zeros(0, np.int64)
    max_coordinate = boxes.max()
    offsets = np.asarray(classes).astype(np.float32) * (max_coordinate + np.float32(1))
    return _nms_plain(boxes + offsets[:, None], scores, None, iou_threshold)


NMS_ROUTES = ("per_class", "coordinate_offset", "torchvision_cpu", "torchvision_gpu")


def batched_nms(boxes, scores, classes, iou_threshold, route="per_class"):
    if route in ("torchvision_cpu", "torchvision_gpu"):
        route = "per_class" if np.asarray(boxes).size > (4000 if route == "torchvision_cpu" else 20000) else "coordinate_offset"
    return (nms_class_aware if route == "per_class" else nms_coordinate_offset)(boxes, scores, classes, iou_threshold)


def post_process(pred_boxes, pred_sims, confidence_threshold=0.75, iou_threshold=0.3, top_k=None, route="per_class"):
    """ref src/models.py:127-146 for ONE image ([P,4], [P,C]) -> (boxes [K,4], classes [K], scores [K], patch_idx [K]).

    ``top_k`` = the eval loop's ``torch.topk(scores, min(200, K))`` (ref main.py:114-117); because the NMS result is
    already score-descending this is a prefix."""
    b = np.asarray(pred_boxes, np.float32)
    s = np.asarray(pred_sims, np.float32)
    scores = s.max(axis=1)
    classes = s.argmax(axis=1)                  # first maximal class, like torch.max on CPU
    sel = np.nonzero(scores > np.float32(confidence_threshold))[0]
    keep = batched_nms(b[sel], scores[sel], classes[sel], iou_threshold, route)
    idx = sel[keep]
    if top_k is not None:
        idx = idx[:top_k]
    return b[idx], classes[idx].astype(np.int64), scores[idx], idx.astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# Input pipeline: ref src/dataset.py:69-71 ``image_processor(images=image,
# return_tensors="pt")["pixel_values"]`` = HF OwlViTImageProcessor (PIL backend): PIL bicubic resize of the u8 RGB image
# to SxS (no crop) -> f32(f64(u8) * (1/255)) -> (x - mean) / std in f32 (transformers image_transforms.rescale /
# normalize).  Pillow (third-party, not under /root/reference; 12.2.0 here) owns the resize arithmetic; its published
# algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
# Vertical_8bpc) is restated: f64 filter weights normalised per output pixel, converted to 22-bit fixed point, a
# horizontal pass rounded/clamped to u8, then a vertical pass.  Pinned against PIL + the HF processor themselves by
# tests/golden/make_golden.py f7 (run in the build container) and against PIL directly in tests when it is importable.
# ---------------------------------------------------------------------------------------------------------------------
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box -> (bounds [out,2] i32 = (first tap,
    tap count), kk [out,ksize] i32 fixed-point weights, ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int) cast truncates toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_pil_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pil_pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One 8bpc resample pass along `axis` (0 = vertical, 1 = horizontal) of an [H,W,C] u8 image."""
    n_in = img.shape[axis]
    ksize = kk.shape[1]
    idx = np.minimum(bounds[:, :1] + np.arange(ksize)[None, :], n_in - 1)       # [out, ksize]; padded taps have k = 0
    src = img.astype(np.int64)
    acc = np.full((img.shape[0] if axis == 1 else len(bounds), len(bounds) if axis == 1 else img.shape[1], img.shape[2]),
                  1 << (_PIL_PRECISION_BITS - 1), np.int64)
    for t in range(ksize):
        if axis == 1:
            acc += src[:, idx[:, t], :] * kk[None, :, t, None]
        else:
            acc += src[idx[:, t], :, :] * kk[:, t, None, None]
    return np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(img_hwc: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL ``Image.resize((out_w, out_h), BICUBIC)`` on an RGB u8 image: horizontal pass, then vertical pass."""
    img = np.ascontiguousarray(img_hwc, dtype=np.uint8)
    H, W = img.shape[:2]
    if W != out_w:
        bx, kx, _ = pil_bicubic_coeffs(W, out_w)
        img = _pil_pass(img, bx, kx, axis=1)
    if H != out_h:
        by, ky, _ = pil_bicubic_coeffs(H, out_h)
        img = _pil_pass(img, by, ky, axis=0)
    return img


CLIP_MEAN_HF = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD_HF = (0.26862954, 0.26130258, 0.27577711)


def normalize_lut(mean=CLIP_MEAN_HF, std=CLIP_STD_HF) -> np.ndarray:
    """[3,256] f32: the value HF's rescale(1/255) + normalize produce for each u8 level (transformers
    image_transforms.py rescale: f64 multiply then f32 cast; normalize: f32 subtract / divide)."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    m = np.array(mean, dtype=np.float32)[:, None]
    s = np.array(std, dtype=np.float32)[:, None]
    return ((v[None, :] - m) / s).astype(np.float32)


def preprocess_image(img_hwc: np.ndarray, size: int, mean=CLIP_MEAN_HF, std=CLIP_STD_HF) -> np.ndarray:
    """ref src/dataset.py:69-71 for one RGB u8 image -> pixel_values [3,size,size] f32."""
    r = pil_resize_bicubic_u8(img_hwc, size, size)
    lut = normalize_lut(mean, std)
    return np.stack([lut[c][r[:, :, c]] for c in range(3)], 0)


# ---------------------------------------------------------------------------------------------------------------------
# Query-bank initialisation: ref src/models.py:155-169 ``_model(**inputs).text_embeds``.
# HF5:603-663 OwlViTTextTransformer (token + position embedding, pre-LN causal encoder with quick-GELU MLPs, final
# LayerNorm, EOS pooling = arg-max token id) -> HF5:952 text_projection (no bias) -> HF5:958 L2 normalise.
# Pinned by fixture F8 (HF itself, run in the build container, incl. the processor-style padding attention_mask).
# ---------------------------------------------------------------------------------------------------------------------
def text_forward(tc, w, input_ids) -> torch.Tensor:
    """tc: TextConfig; w: name -> array (weights.text_param_shapes names); input_ids [N,S] -> text_embeds [N,proj] f32."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.int64)
    N, S = ids.shape
    g = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in w.items()}
    x = g["text_model.embeddings.token_embedding.weight"][ids] + g["text_model.embeddings.position_embedding.weight"][:S][None]
    H, dh = tc.heads, tc.width // tc.heads
    causal = torch.full((S, S), float("-inf")).triu(1)
    for i in range(tc.layers):
        pre = f"text_model.encoder.layers.{i}."
        h = F.layer_norm(x, (tc.width,), g[pre + "layer_norm1.weight"], g[pre + "layer_norm1.bias"], tc.ln_eps)
        q = F.linear(h, g[pre + "self_attn.q_proj.weight"], g[pre + "self_attn.q_proj.bias"]).view(N, S, H, dh).transpose(1, 2)
        k = F.linear(h, g[pre + "self_attn.k_proj.weight"], g[pre + "self_attn.k_proj.bias"]).view(N, S, H, dh).transpose(1, 2)
        v = F.linear(h, g[pre + "self_attn.v_proj.weight"], g[pre + "self_attn.v_proj.bias"]).view(N, S, H, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5 + causal, dim=-1) @ v
        a = a.transpose(1, 2).reshape(N, S, tc.width)
        x = x + F.linear(a, g[pre + "self_attn.out_proj.weight"], g[pre + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (tc.width,), g[pre + "layer_norm2.weight"], g[pre + "layer_norm2.bias"], tc.ln_eps)
        u = quick_gelu(F.linear(h, g[pre + "mlp.fc1.weight"], g[pre + "mlp.fc1.bias"]))
        x = x + F.linear(u, g[pre + "mlp.fc2.weight"], g[pre + "mlp.fc2.bias"])
    x = F.layer_norm(x, (tc.width,), g["text_model.final_layer_norm.weight"], g["text_model.final_layer_norm.bias"], tc.ln_eps)
    pooled = x[torch.arange(N), ids.argmax(dim=-1)]
    e = F.linear(pooled, g["text_projection.weight"])
    return e / torch.linalg.norm(e, ord=2, dim=-1, keepdim=True)
