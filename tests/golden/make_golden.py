"""Generate the golden fixtures under tests/golden by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and `transformers`); never on the GPU box.
What is committed are the OUTPUT VECTORS (.npz) plus this script -- no reference source.

Recipe:
  * stub `torchvision` (absent here) -- only `box_area` is used on the train path;
  * build HF `OwlViTForObjectDetection` from a config (no hub access), eager attention;
  * load this repo's deterministic weights (weights.make_weights) by name mapping;
  * wrap with the reference's `src.models.OwlViT`, re-apply its freeze loop
    (src/models.py:173-184 lives inside `load_model`, which needs the network);
  * adapt `compute_box_bias` (transformers 5.x signature drift; numerically identical);
  * re-enact main.py:74-91 around `src.losses.PushPullLoss` at the reference's batch size of 1.

Usage:  python tests/golden/make_golden.py [f1 f2 f3 f4 f5 f6 f7 f8 f9 f10 lsap  f2b f4b f10b f2c]
        (f2b / f4b / f10b / f2c read the reference outputs stored in f2 / f4 / f10 and must run after them; `f2c` with seed 0 re-runs the 20 000-seed search)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import transformers  # noqa: E402  (must be imported BEFORE the torchvision stub)
from transformers import OwlViTConfig, OwlViTForObjectDetection  # noqa: E402

tv = types.ModuleType("torchvision")
tv_ops = types.ModuleType("torchvision.ops")
tv_ops.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
tv_ops.nms = None


_NMS_ROUTE = ["per_class"]        # f6 runs the reference once per torchvision route


def _stub_batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms is absent here; its published algorithm (torchvision/ops/boxes.py) restated in plain
    torch so that the reference's PostProcess (src/models.py:122-146) can run for fixture F6 -- both routes torchvision
    has: `_batched_nms_vanilla` (per class: greedy NMS by descending score on raw coordinates, suppress IoU > thr) and
    `_batched_nms_coordinate_trick` (boxes + class * (boxes.max() + 1), one class-agnostic NMS); result ordered by
    descending score.  The NMS arithmetic in F6 is therefore NOT torchvision's own ("parity unpinned" for that
    dependency); what F6 pins is everything the reference itself does around it: the max / threshold / indexing /
    ordering / output shapes -- and, between the two routes, how many keep decisions the shifted coordinates flip."""
    if _NMS_ROUTE[0] == "coordinate_offset" and boxes.numel() > 0:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes = boxes + offsets[:, None]
        idxs = torch.zeros_like(idxs)
    n = boxes.shape[0]
    keep_mask = torch.zeros(n, dtype=torch.bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for cls in torch.unique(idxs):
        ids = torch.nonzero(idxs == cls).squeeze(1)
        order = ids[torch.sort(scores[ids], descending=True, stable=True).indices]
        alive = torch.ones(len(order), dtype=torch.bool)
        for a in range(len(order)):
            if not alive[a]:
                continue
            i = order[a]
            keep_mask[i] = True
            rest = order[a + 1:]
            if len(rest) == 0:
                break
            w = (torch.minimum(boxes[i, 2], boxes[rest, 2]) - torch.maximum(boxes[i, 0], boxes[rest, 0])).clamp(min=0)
            h = (torch.minimum(boxes[i, 3], boxes[rest, 3]) - torch.maximum(boxes[i, 1], boxes[rest, 1])).clamp(min=0)
            inter = w * h
            ovr = inter / (area[i] + area[rest] - inter)
            alive[a + 1:] &= ~(ovr > iou_threshold)
    keep = torch.nonzero(keep_mask).squeeze(1)
    return keep[torch.sort(scores[keep], descending=True, stable=True).indices]


tv_ops.batched_nms = _stub_batched_nms
tv.ops = tv_ops
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tv_ops

sys.path.insert(0, "/root/reference")
from src.models import OwlViT as RefOwlViT  # noqa: E402
from src.losses import PushPullLoss as RefPushPullLoss  # noqa: E402

from owl_vit_object_detection_amd import synth, weights  # noqa: E402
from owl_vit_object_detection_amd.config import get_config  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_reference_model(cfg, seed=1234, profile="init"):
    hf_cfg = OwlViTConfig(
        vision_config=dict(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                           num_attention_heads=cfg.heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                           hidden_act="quick_gelu", layer_norm_eps=cfg.ln_eps),
        text_config=dict(hidden_size=cfg.text_dim, intermediate_size=64, num_hidden_layers=1,
                         num_attention_heads=1, vocab_size=64, max_position_embeddings=16),
        projection_dim=cfg.text_dim,
    )
    hf_cfg._attn_implementation = "eager"
    hf_cfg.vision_config._attn_implementation = "eager"
    hf = OwlViTForObjectDetection(hf_cfg)
    W = weights.make_weights(cfg, seed, profile)
    sd = hf.state_dict()
    for name, arr in W.items():
        if name == "queries":
            continue
        if name.startswith("backbone."):
            key = "owlvit.vision_model." + name[len("backbone."):]
        elif name.startswith("post_post_layernorm."):
            key = "layer_norm." + name[len("post_post_layernorm."):]
        elif name.startswith("class_predictor.dense0."):
            key = "class_head.dense0." + name[len("class_predictor.dense0."):]
        else:
            key = name
        assert key in sd and tuple(sd[key].shape) == arr.shape, (name, key)
        sd[key] = torch.from_numpy(arr.copy())
    hf.load_state_dict(sd)
    model = RefOwlViT(pretrained_model=hf, query_bank=torch.from_numpy(W["queries"].copy()))
    model.compute_box_bias = lambda fm: hf.compute_box_bias(fm.shape[1], fm.shape[2])
    # freeze loop: reference src/models.py:173-184
    for name, parameter in model.named_parameters():
        conditions = ["layers.11" in name, "box" in name, "post_layernorm" in name,
                      "class_predictor" in name, "queries" in name]
        if any(conditions):
            continue
        parameter.requires_grad = False
    names = [n for n, _ in model.named_parameters()]
    assert set(names) == set(weights.param_shapes(cfg).keys()), set(names) ^ set(weights.param_shapes(cfg).keys())
    for n, p in model.named_parameters():
        assert p.requires_grad == weights.is_trainable(n), n
    model.train()
    return model, W


def run_reference_step(model, cfg, image, labels, boxes, scales, taps=None):
    """main.py:74-91 re-enacted at batch 1. image [1,3,S,S]; labels [n]; boxes [n,4]."""
    hooks = []
    if taps is not None:
        bb = model.backbone
        hooks.append(bb.embeddings.register_forward_hook(lambda m, i, o: taps.__setitem__("embed", o.detach().clone())))
        hooks.append(bb.pre_layernorm.register_forward_hook(lambda m, i, o: taps.__setitem__("pre_ln", o.detach().clone())))
        for li, layer in enumerate(bb.encoder.layers):
            hooks.append(layer.register_forward_hook(
                lambda m, i, o, li=li: taps.__setitem__(f"backbone.encoder.layers.{li}.out",
                                                        (o[0] if isinstance(o, tuple) else o).detach().clone())))
        hooks.append(model.post_post_layernorm.register_forward_hook(lambda m, i, o: taps.__setitem__("feats", o.detach().clone())))
    for p in model.parameters():
        p.grad = None
    criterion = RefPushPullLoss(cfg.n_classes, scales=None if scales is None else torch.tensor(scales))
    pred_boxes, _none1, pred_sims, _none2 = model(torch.from_numpy(image))
    assert _none1 is None and _none2 is None
    lab = torch.from_numpy(labels)[None]
    tb = torch.from_numpy(boxes)[None]
    # capture matcher output (pre-spreading) by calling the matcher the same way forward() does
    with torch.no_grad():
        tc0, indices, _ = criterion.matcher({"pred_logits": pred_sims, "pred_boxes": pred_boxes},
                                            [{"labels": lab[0], "boxes": tb[0]}])
    losses = criterion(pred_sims, lab, pred_boxes, tb)
    loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
    loss.backward()
    for h in hooks:
        h.remove()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    out = dict(pred_boxes=pred_boxes.detach().numpy(), pred_sims=pred_sims.detach().numpy(),
               target_classes_matched=tc0[0].numpy(), pred_idx=indices[0][0].numpy(), tgt_idx=indices[0][1].numpy())
    for k, v in losses.items():
        out[k] = np.float32(v.item())
    return out, grads


def recover_spread_labels(cfg, out):
    """target_classes AFTER spreading is not returned by the reference; re-run its exact loop
    (src/losses.py:100-106) on the captured tensors with the reference's own box_iou."""
    from src.matcher import box_iou
    pb = torch.from_numpy(out["pred_boxes"])
    tc = torch.from_numpy(out["target_classes_matched"].copy())[None]
    for box, label in zip(pb[0], tc[0]):
        if label == cfg.n_classes:
            continue
        iou, _ = box_iou(box.unsqueeze(0), pb.squeeze(0))
        idx = iou > 0.85
        tc[idx] = label.item()
    return tc[0].numpy()


def grad_summary(grads, full):
    d = {}
    for n, g in grads.items():
        g = g.numpy()
        if full:
            d["grad/" + n] = g
        else:
            d["gradnorm/" + n] = np.float64(np.linalg.norm(g.astype(np.float64)))
            d["gradhead/" + n] = g.reshape(-1)[:64].copy()
            # a strided 4096-element sample over the WHOLE tensor (a 64-element head is bf16 noise for the small tensors)
            flat = g.reshape(-1)
            stride = max(1, flat.size // 4096)
            d["gradsample/" + n] = flat[::stride][:4096].copy()
    return d


def f1():
    for cname in ("tiny", "tiny-l14"):
        cfg = get_config(cname)
        model, _ = build_reference_model(cfg)
        img = synth.make_images(cfg, 1)
        labels, boxes = synth.make_targets(cfg, 1, max_boxes=6)
        scales = synth.class_scales(cfg, labels)
        taps = {}
        out, grads = run_reference_step(model, cfg, img, labels[0], boxes[0], scales, taps)
        out["target_classes"] = recover_spread_labels(cfg, out)
        out["scales"] = scales
        for k, v in taps.items():
            out["tap/" + k] = v.numpy()
        out.update(grad_summary(grads, full=True))
        np.savez_compressed(os.path.join(HERE, f"f1_{cname}.npz"), **out)
        print("f1", cname, {k: float(out[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")})


def _full(cname, tag, seed=1234):
    cfg = get_config(cname)
    model, _ = build_reference_model(cfg, seed)
    img = synth.make_images(cfg, 1, seed)
    labels, boxes = synth.make_targets(cfg, 1, seed, max_boxes=16)
    scales = synth.class_scales(cfg, labels)
    out, grads = run_reference_step(model, cfg, img, labels[0], boxes[0], scales)
    out["target_classes"] = recover_spread_labels(cfg, out)
    out["scales"] = scales
    out["pred_boxes"] = out["pred_boxes"].astype(np.float32)
    out.update(grad_summary(grads, full=False))
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print(tag, {k: float(out[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")})


def f2():
    _full("owlvit-base-patch16", "f2_b16")


# ---------------------------------------------------------------------------------------------------
# F2b / F4b / F10b (+ F2c): full-size fixtures whose every DISCRETE decision and every kink of the loss has margin.
#
# F2 / F4 / F10 draw their targets from synth.make_targets.  On every one of them a matched coordinate sits within 3e-3 of its target (F2 1.6e-4, F4 7.7e-5,
# F10 5.2e-4): sign(pred - tgt) of the L1 term (ref src/losses.py:57) flips under the bf16 forward's ~2e-3 deviation, that row carries the image's largest box
# gradient, and the end-to-end gradient comparison of every box-fed tensor is then loose / skipped.  This is the expected case, not bad luck: at HF-init weights
# a seeded target (0.02-0.37 wide) is 2-20x wider than the predictions (0.01-0.13), every prediction that lies inside a target has the SAME L1 cost (tw - pw) and
# nearly the same GIoU, so the assignment's runner-up gap is head noise and the winner hugs an edge: over the 11 376 of 20 000 seeds with n >= 8 (`search_seeds`) the gap has median
# 1.1e-3 (max 3.5e-2), 76 seeds have a coordinate margin >= 5e-3, 26 pass criteria 1, 2, 4, 5 -- and the best runner-up gap among those is 6.2e-3 (seed 2347, n = 8).  So:
#   * F2c = the best a SEED search reaches (criteria 1, 2, 4 hold, runner-up gap as large as the search finds; `search_seeds` below, margins stored in the file);
#   * F2b / F4b / F10b = targets CONSTRUCTED around predictions of the reference itself (`anchored_targets`): n anchor rows spread over the grid, target =
#     the anchor's predicted box with every edge moved by 25-45 % of the box's extent (>= 6e-3; patterns grow / shift left / shift right per axis, never shrink),
#     label drawn until |sim[anchor, label]| >= 2e-2.  All four criteria then hold with room, and `decision_margins` VERIFIES them on the reference's own outputs
#     before anything is written:
#       1. every matched coordinate >= 5e-3 from its target                              (L1 kink, ref src/losses.py:57)
#       2. every min / max / clamp selection of the matched pairs' GIoU has >= 5e-3      (ref src/matcher.py:25-44: intersection / hull corners select between a
#          prediction's and its target's coordinate -- criterion 1 -- and the intersection's clamp(min=0) needs |extent| >= 5e-3)
#       3. runner-up assignment (each matched pair forbidden in turn, scipy re-solved) costs >= 2.5e-2 more = 10x the measured forward error (2.1e-3 ... 2.4e-3)
#       4. no |IoU - 0.85| < 2e-2 between a positive row and any row in the spreading scan (ref src/losses.py:100-106)
#       5. (added) |sim| of every positive row at its label >= 2e-2: the class term's slope is -w / |sim| (ref src/losses.py:21,34)
# ---------------------------------------------------------------------------------------------------
def decision_margins(cfg, pb, ps, labels, boxes):
    """Margins of the reference's outputs (pb [P,4], ps [P,C]) for targets (labels [n], boxes [n,4]) against the five criteria above, with the reference's own
    matcher arithmetic (src/matcher.py) and scipy."""
    from scipy.optimize import linear_sum_assignment
    from src.matcher import box_iou, generalized_box_iou
    pbt, pst = torch.from_numpy(pb), torch.from_numpy(ps)
    lab, tgt = torch.from_numpy(labels).long(), torch.from_numpy(boxes).float()
    C = (torch.cdist(pbt, tgt, p=1) - pst.softmax(-1)[:, lab] - generalized_box_iou(pbt, tgt)).double().numpy()     # src/matcher.py:106-124, unit weights
    r, c = linear_sum_assignment(C)
    best = C[r, c].sum()
    gap = np.inf
    for k in range(len(r)):
        C2 = C.copy(); C2[r[k], c[k]] = 1e9
        r2, c2 = linear_sum_assignment(C2)
        gap = min(gap, C2[r2, c2].sum() - best)
    src, dst = pb[r], boxes[c]
    coord = float(np.abs(src - dst).min())
    iw = np.minimum(src[:, 2], dst[:, 2]) - np.maximum(src[:, 0], dst[:, 0])
    ih = np.minimum(src[:, 3], dst[:, 3]) - np.maximum(src[:, 1], dst[:, 1])
    inter = float(min(np.abs(iw).min(), np.abs(ih).min()))
    tc = torch.full((pb.shape[0],), cfg.n_classes, dtype=torch.long)
    tc[torch.from_numpy(r)] = lab[torch.from_numpy(c)]
    iou_margin = 1.0
    for p in range(pb.shape[0]):                                   # the spreading scan (src/losses.py:100-106), margins taken on the way
        if tc[p] == cfg.n_classes:
            continue
        iou = box_iou(pbt[p:p + 1], pbt)[0][0]
        iou_margin = min(iou_margin, float((iou - 0.85).abs().min()))
        tc[iou > 0.85] = tc[p]
    pos = torch.nonzero(tc != cfg.n_classes).flatten()
    return dict(n=int(len(r)), gap=float(gap), coord=coord, inter=inter, iou=iou_margin, simpos=float(pst[pos, tc[pos]].abs().min()), npos=int(len(pos)),
                pred_idx=r[np.argsort(c)])


MARGIN_BARS = dict(coord=5e-3, inter=5e-3, gap=2.5e-2, iou=2e-2, simpos=2e-2)


def anchored_targets(cfg, pb, ps, seed, n=12, column_margin=4e-2):
    """n targets constructed around predictions of the reference (see the block comment above).  Deterministic in (seed, reference outputs).  A candidate
    anchor is kept only if its own prediction wins its target's cost column by `column_margin` (another, larger prediction nearby may fit the moved box
    better): the assignment is then the column-wise argmin and its runner-up gap is at least that margin."""
    from owl_vit_object_detection_amd import rng
    from src.matcher import generalized_box_iou
    G = cfg.grid
    w, h = pb[:, 2] - pb[:, 0], pb[:, 3] - pb[:, 1]
    pbt, prob = torch.from_numpy(pb), torch.from_numpy(ps).softmax(-1)
    order = np.argsort(rng.uniform(seed, "anchor/order", cfg.patches))
    anchors, labels, boxes = [], [], []
    for p in order:
        p = int(p)
        py, px = divmod(p, G)
        if not (2 <= px < G - 2 and 2 <= py < G - 2) or min(w[p], h[p]) < 0.0125:
            continue
        if any(max(abs(py - qy), abs(px - qx)) < 6 for qy, qx in (divmod(q, G) for q in anchors)):
            continue
        u = rng.uniform(seed, f"anchor/{p}", 8)
        d = np.zeros(4)
        for ax, ext in ((0, w[p]), (1, h[p])):
            m0 = np.clip((0.25 + 0.2 * u[2 * ax]) * ext, 6e-3, ext - 6e-3)
            m1 = np.clip((0.25 + 0.2 * u[2 * ax + 1]) * ext, 6e-3, ext - 6e-3)
            pat = int(u[4 + ax] * 3)                      # 0 grow, 1 shift towards +, 2 shift towards -
            d[ax], d[ax + 2] = ((-m0, m1), (m0, m1), (-m0, -m1))[pat]
        box = (pb[p].astype(np.float64) + d).astype(np.float32)
        cand = [int(c) for c in rng.randint(seed, f"anchor/label/{p}", 32, cfg.n_classes) if abs(ps[p, c]) >= 2.5e-2]
        if not cand:
            continue
        t = torch.from_numpy(box)[None]
        col = (torch.cdist(pbt, t, p=1) - prob[:, cand[0]:cand[0] + 1] - generalized_box_iou(pbt, t))[:, 0].double().numpy()
        others = np.delete(col, p)
        if col[p] + column_margin > others.min():
            continue
        anchors.append(p); labels.append(cand[0]); boxes.append(box)
        if len(anchors) == n:
            break
    assert len(anchors) == n, len(anchors)
    return np.array(labels, np.int64), np.stack(boxes).astype(np.float32), np.array(anchors, np.int64)


def _check_margins(m, tag, bars=MARGIN_BARS):
    print(tag, "decision margins:", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in m.items() if k != "pred_idx"})
    for k, bar in bars.items():
        assert m[k] >= bar, (tag, k, m[k], bar)


def _full_margins(cname, src_tag, tag, profile="init", n=12, targets=None, bars=MARGIN_BARS, seed=1234):
    """One reference step at batch 1 on targets with margins; the targets themselves are stored (tgt_labels / tgt_boxes): the GPU-side test needs neither
    the reference nor this generator."""
    cfg = get_config(cname)
    src = np.load(os.path.join(HERE, f"{src_tag}.npz"))
    pb, ps = src["pred_boxes"][0], src["pred_sims"][0]
    if targets is None:
        labels, boxes, anchors = anchored_targets(cfg, pb, ps, seed=77, n=n)
    else:
        labels, boxes = targets
        anchors = None
    m = decision_margins(cfg, pb, ps, labels, boxes)
    _check_margins(m, tag, bars)
    if anchors is not None:
        assert sorted(m["pred_idx"].tolist()) == sorted(anchors.tolist()), "an anchor lost its own target"
    model, _ = build_reference_model(cfg, seed, profile=profile)
    img = synth.make_images(cfg, 1, seed)
    scales = synth.class_scales(cfg, [labels])
    out, grads = run_reference_step(model, cfg, img, labels, boxes, scales)
    assert np.array_equal(out["pred_boxes"], src["pred_boxes"]) and np.array_equal(out["pred_sims"], src["pred_sims"]), "the source fixture is not this model's output"
    m2 = decision_margins(cfg, out["pred_boxes"][0], out["pred_sims"][0], labels, boxes)
    assert np.array_equal(np.sort(m2["pred_idx"]), np.sort(out["pred_idx"]))
    out["target_classes"] = recover_spread_labels(cfg, out)
    out["scales"] = scales
    out["tgt_labels"], out["tgt_boxes"] = labels, boxes
    for k in ("gap", "coord", "inter", "iou", "simpos"):
        out["margin/" + k] = np.float64(m[k])
    out["pred_boxes"] = out["pred_boxes"].astype(np.float32)
    out.update(grad_summary(grads, full=False))
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print(tag, {k: float(out[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")}, "positives after spreading", int((out["target_classes"] != cfg.n_classes).sum()))


def f2b():
    _full_margins("owlvit-base-patch16", "f2_b16", "f2b_b16_margins")


def f4b():
    _full_margins("owlvit-large-patch14", "f4_l14", "f4b_l14_margins")


def f10b():
    _full_margins("owlvit-base-patch16", "f10_b16_trained", "f10b_b16_trained_margins", profile="trained_like")


def search_seeds(cname="owlvit-base-patch16", src_tag="f2_b16", n_seeds=20000, min_boxes=8):
    """The seed search VERDICT r04 #1 asks for, over synth.make_targets' seed on the reference's stored outputs: criteria 1, 2, 4, 5 as bars, criterion 3
    (runner-up gap) maximised.  Prints the distribution so that the docstring above can be checked."""
    cfg = get_config(cname)
    src = np.load(os.path.join(HERE, f"{src_tag}.npz"))
    pb, ps = src["pred_boxes"][0], src["pred_sims"][0]
    rows = []
    for seed in range(1, n_seeds + 1):
        labels, boxes = synth.make_targets(cfg, 1, seed, max_boxes=16)
        if len(labels[0]) < min_boxes:
            continue
        m = decision_margins(cfg, pb, ps, labels[0], boxes[0])
        rows.append((m["gap"], m["coord"], m["inter"], m["iou"], m["simpos"], seed, m["n"]))
    gaps = np.array([r[0] for r in rows])
    ok = [r for r in rows if r[1] >= 5e-3 and r[2] >= 5e-3 and r[3] >= 2e-2 and r[4] >= 1e-2]
    print(f"search_seeds: {len(rows)} seeds with n >= {min_boxes}: runner-up gap median {np.median(gaps):.2e} max {gaps.max():.2e}; "
          f"{sum(r[1] >= 5e-3 for r in rows)} with coordinate margin >= 5e-3; {len(ok)} pass criteria 1, 2, 4, 5")
    ok.sort(reverse=True)
    print("   best by gap:", ok[:5])
    return ok[0][5] if ok else None


F2C_SEED = 2347        # = search_seeds()'s result (20 000 seeds, 4.5 minutes); f2c(seed=0) re-runs the search


def f2c(seed=None):
    cfg = get_config("owlvit-base-patch16")
    seed = seed if seed else (search_seeds() if seed == 0 else F2C_SEED)            # seed=0: search again; None: the recorded result
    labels, boxes = synth.make_targets(cfg, 1, seed, max_boxes=16)
    bars = dict(MARGIN_BARS, gap=0.0, simpos=1e-2)
    _full_margins("owlvit-base-patch16", "f2_b16", "f2c_b16_seed_search", targets=(labels[0], boxes[0]), bars=bars)
    g = dict(np.load(os.path.join(HERE, "f2c_b16_seed_search.npz")))
    g["target_seed"] = np.int64(seed)
    np.savez_compressed(os.path.join(HERE, "f2c_b16_seed_search.npz"), **g)


def attention_stats(model, cfg, image):
    """Statistics of the reference's own attention logits on `image` (hooks on every layer's layer_norm1): per layer the std / max of the logits
    and a simulation of the HIP forward's softmax-offset logic on them -- the kernel (csrc/attention_fwd.hip) exponentiates a 64-key tile against
    the offset it already holds and only recomputes (`slow path`) when a row sum of the tile exceeds 2^88 (csrc/attention_fwd_common.h); one wave = 32 consecutive queries, the class
    token is peeled (key 0 is the initial state, |s0| > 88 sets the offset), tiles cover keys 1..T-1 in order.  `slow_tiles` = (wave, tile) pairs
    that would take the slow path at this layer, all heads (the first tile of a wave does not count: it cannot overflow an empty state)."""
    hs = {}
    hooks = []
    for li, layer in enumerate(model.backbone.encoder.layers):
        hooks.append(layer.layer_norm1.register_forward_hook(lambda m, i, o, li=li: hs.__setitem__(li, o.detach())))
    with torch.no_grad():
        model(torch.from_numpy(image))
    for h in hooks:
        h.remove()
    H, dh, T = cfg.heads, cfg.head_dim, cfg.tokens
    LOG2E = 1.4426950408889634
    std, mx, slow, peak = [], [], [], []
    for li, layer in enumerate(model.backbone.encoder.layers):
        h = hs[li][0]
        sa = layer.self_attn
        q = (h @ sa.q_proj.weight.T + sa.q_proj.bias).view(T, H, dh).transpose(0, 1)
        k = (h @ sa.k_proj.weight.T + sa.k_proj.bias).view(T, H, dh).transpose(0, 1)
        att = torch.matmul(q, k.transpose(1, 2)) * (dh ** -0.5)            # [H, T, T] natural-log units
        std.append(float(att.std())); mx.append(float(att.abs().max()))
        peak.append(float(torch.softmax(att, -1).max(-1).values.mean()))
        s2 = att[:, 1:, :] * LOG2E                                          # queries 1..T-1 (query 0 is the VALU-only workgroup)
        nq = (T - 1) // 32 * 32
        n_slow = 0
        s0 = s2[:, :nq, 0]
        M = torch.where(s0.abs() > 88, s0, torch.zeros_like(s0))           # [H, nq] per-query offset
        keys = s2[:, :nq, 1:]
        for t0 in range(0, keys.shape[-1] - 63, 64):
            tile = keys[:, :, t0:t0 + 64]
            rs = torch.exp2((tile - M[..., None]).double()).sum(-1)        # row sums against the held offset
            trip = (rs > 2.0 ** 88).view(H, nq // 32, 32).any(-1)          # wave-uniform verdict
            n_slow += int(trip.sum())
            tm = tile.max(-1).values
            upd = trip[..., None].expand(H, nq // 32, 32).reshape(H, nq)
            M = torch.where(upd, torch.maximum(M, tm), M)
        slow.append(n_slow)
    return dict(logit_std=np.array(std), logit_absmax=np.array(mx), softmax_peak_mean=np.array(peak), slow_tiles=np.array(slow, np.int64))


def f10():
    """F10 -- trained-like statistics end to end: weights.make_weights(cfg, profile="trained_like") through the reference at batch 1,
    tiny (all intermediates + all gradients) and full B/16 768^2 (outputs, losses, decisions, gradient norms + samples), plus the attention statistics
    of the reference's own logits (what the HIP attention forward's offset / verdict logic will meet)."""
    for cname, tag, full, profile in (("tiny", "f10_tiny_trained", True, "trained_like"), ("owlvit-base-patch16", "f10_b16_trained", False, "trained_like"),
                                      ("owlvit-base-patch16", "f10_b16_trained_hard", False, "trained_like_hard")):
        cfg = get_config(cname)
        model, _ = build_reference_model(cfg, profile=profile)
        img = synth.make_images(cfg, 1)
        labels, boxes = synth.make_targets(cfg, 1, max_boxes=6 if full else 16)
        scales = synth.class_scales(cfg, labels)
        taps = {} if full else None
        out, grads = run_reference_step(model, cfg, img, labels[0], boxes[0], scales, taps)
        out["target_classes"] = recover_spread_labels(cfg, out)
        out["scales"] = scales
        if full:
            for k, v in taps.items():
                out["tap/" + k] = v.numpy()
        out.update(grad_summary(grads, full=full))
        st = attention_stats(model, cfg, img)
        for k, v in st.items():
            out["attn/" + k] = v
        np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
        print(tag, {k: float(out[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")})
        print("   |sims| max", float(np.abs(out["pred_sims"]).max()), "quantiles", np.quantile(np.abs(out["pred_sims"]), [0.5, 0.9, 0.99]).round(3),
              "boxes range", float(out["pred_boxes"].min()), float(out["pred_boxes"].max()))
        print("   logit std per layer", st["logit_std"].round(2), "\n   |logit| max", st["logit_absmax"].round(1),
              "\n   mean softmax peak", st["softmax_peak_mean"].round(3), "\n   slow tiles per layer", st["slow_tiles"])


def f3():
    """Batched semantics: B separate batch-1 reference runs, losses and grads averaged."""
    cfg = get_config("small")
    B = 3
    model, _ = build_reference_model(cfg)
    imgs = synth.make_images(cfg, B)
    labels, boxes = synth.make_targets(cfg, B, max_boxes=8)
    scales = synth.class_scales(cfg, labels)
    acc, gacc = None, None
    pb, ps, tcs = [], [], []
    for b in range(B):
        out, grads = run_reference_step(model, cfg, imgs[b:b + 1], labels[b], boxes[b], scales)
        pb.append(out["pred_boxes"][0]); ps.append(out["pred_sims"][0])
        tcs.append(recover_spread_labels(cfg, out))
        l = {k: float(out[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")}
        acc = l if acc is None else {k: acc[k] + l[k] for k in acc}
        gacc = grads if gacc is None else {k: gacc[k] + grads[k] for k in gacc}
    res = dict(pred_boxes=np.stack(pb), pred_sims=np.stack(ps), target_classes=np.stack(tcs), scales=scales)
    for k, v in acc.items():
        res[k] = np.float32(v / B)
    res.update(grad_summary({k: v / B for k, v in gacc.items()}, full=False))
    np.savez_compressed(os.path.join(HERE, "f3_small_batch3.npz"), **res)
    print("f3", {k: float(res[k]) for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")})


def f4():
    _full("owlvit-large-patch14", "f4_l14")


def f5():
    """Loss-only adversarial cases straight through the reference PushPullLoss (no model)."""
    from owl_vit_object_detection_amd import rng
    cases = {}

    def boxes_from(seed, stream, n, wmin=0.02, wmax=0.37):
        x0 = rng.uniform(seed, stream, n, 0) * 0.6
        y0 = rng.uniform(seed, stream, n, 1) * 0.6
        w = wmin + rng.uniform(seed, stream, n, 2) * (wmax - wmin)
        h = wmin + rng.uniform(seed, stream, n, 3) * (wmax - wmin)
        return np.stack([x0, y0, x0 + w, y0 + h], 1).astype(np.float32)

    def add(name, P, C, n, scales, dup_chain=False, ties=False, sim_one=False):
        sims = (rng.uniform(7, name + "/sims", P * C).reshape(P, C) * 1.6 - 0.8).astype(np.float32)
        pb = boxes_from(7, name + "/pb", P, 0.05, 0.3)
        tb = boxes_from(7, name + "/tb", n)
        labels = rng.randint(7, name + "/lab", n, C)
        if dup_chain:
            # chains of near-duplicates: p+1 is p shifted by a hair -> IoU > 0.85 with p, lower with p-2
            for s in range(0, P - 12, 12):
                for k in range(1, 10):
                    pb[s + k] = pb[s] + np.float32(0.012 * k) * np.array([1, 0, 1, 0], np.float32)
            # and make some chain heads sit exactly on targets so they get matched
            for t in range(min(n, P // 12)):
                pb[12 * t] = tb[t]
                for k in range(1, 10):
                    pb[12 * t + k] = tb[t] + np.float32(0.012 * k) * np.array([1, 0, 1, 0], np.float32)
        if ties:
            pb[1::2] = pb[0::2][: len(pb[1::2])]
            sims[1::2] = sims[0::2][: len(sims[1::2])]
        if sim_one:
            sims[3, 1] = 1.0
            sims[5, 0] = -1.0
            sims[6, 2] = 0.0
        cases[name] = (sims, pb, labels, tb, scales)

    add("basic_n7_noscale", 200, 10, 7, None)
    add("basic_n7_scale", 200, 10, 7, np.round(3 + rng.uniform(7, "sc", 10) * 2, 1).astype(np.float32))
    add("n1", 64, 10, 1, None)
    add("n40", 300, 10, 40, np.round(3 + rng.uniform(7, "sc2", 10) * 2, 1).astype(np.float32))
    add("dup_chain", 240, 10, 6, None, dup_chain=True)
    add("ties", 128, 4, 5, None, ties=True)
    add("sim_one", 96, 4, 3, np.array([3.0, 3.5, 4.1, 3.2], np.float32), sim_one=True)

    res = {}
    for name, (sims, pb, labels, tb, scales) in cases.items():
        C = sims.shape[1]
        crit = RefPushPullLoss(C, scales=None if scales is None else torch.tensor(scales))
        s = torch.from_numpy(sims)[None].clone().requires_grad_(True)
        b = torch.from_numpy(pb)[None].clone().requires_grad_(True)
        lab = torch.from_numpy(labels)[None]
        t = torch.from_numpy(tb)[None]
        with torch.no_grad():
            tc0, indices, _ = crit.matcher({"pred_logits": s, "pred_boxes": b}, [{"labels": lab[0], "boxes": t[0]}])
        losses = crit(s, lab, b, t)
        (losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]).backward()
        out = dict(pred_boxes=pb[None], target_classes_matched=tc0[0].numpy())
        res[name + "/sims"] = sims
        res[name + "/pred_boxes"] = pb
        res[name + "/labels"] = labels
        res[name + "/tgt_boxes"] = tb
        if scales is not None:
            res[name + "/scales"] = scales
        res[name + "/target_classes_matched"] = tc0[0].numpy()
        res[name + "/target_classes"] = recover_spread_labels(types.SimpleNamespace(n_classes=C), out)
        res[name + "/pred_idx"] = indices[0][0].numpy()
        res[name + "/tgt_idx"] = indices[0][1].numpy()
        for k, v in losses.items():
            res[name + "/" + k] = np.float32(v.item())
        res[name + "/grad_sims"] = s.grad[0].numpy()
        res[name + "/grad_boxes"] = b.grad[0].numpy()
        print("f5", name, {k: float(v) for k, v in losses.items()},
              "spread:", int((res[name + "/target_classes"] != C).sum()), "matched:", len(labels))
    np.savez_compressed(os.path.join(HERE, "f5_loss_cases.npz"), **res)


def f6():
    """Post-process cases through the reference's PostProcess (src/models.py:122-146) at its batch size of 1."""
    from src.models import PostProcess as RefPostProcess
    res = {}
    cases = []
    rng_ = np.random.default_rng(77)

    def synth_case(P, C, cluster):
        if cluster:
            centers = rng_.random((8, 2)).astype(np.float32) * 0.6 + 0.1
            c = centers[rng_.integers(0, 8, P)] + rng_.normal(0, 0.01, (P, 2)).astype(np.float32)
            wh = np.float32(0.2) + rng_.normal(0, 0.01, (P, 2)).astype(np.float32)
        else:
            c = rng_.random((P, 2)).astype(np.float32) * 0.7
            wh = rng_.random((P, 2)).astype(np.float32) * 0.3 + np.float32(0.02)
        boxes = np.concatenate([c, c + wh], 1).astype(np.float32)
        sims = ((rng_.random((P, C)).astype(np.float32) * 2 - 1) * np.float32(0.6)).astype(np.float32)
        return boxes, sims

    cases.append(synth_case(576, 10, False) + (0.01, 0.6))       # config.yaml:13-14 thresholds
    cases.append(synth_case(2304, 10, True) + (0.3, 0.45))
    cases.append(synth_case(2304, 10, False) + (0.01, 0.6))
    cases.append(synth_case(144, 4, True) + (0.75, 0.3))         # the class defaults (models.py:123): nothing passes
    # a real model output (tiny config through the reference model)
    cfg = get_config("tiny")
    model, _ = build_reference_model(cfg)
    model.eval()
    with torch.no_grad():
        pb, _, ps, _ = model(torch.from_numpy(synth.make_images(cfg, 1, 3)))
    cases.append((pb[0].numpy().copy(), ps[0].numpy().copy(), 0.01, 0.6))
    # near-ties for the coordinate-offset route: pairs of same-class boxes whose IoU sits within a few f32 ulps of the threshold
    nt_b, nt_s = synth_case(2304, 10, False)
    thr = np.float32(0.6)
    for q in range(0, 2300, 2):
        x1, y1, x2, y2 = nt_b[q]
        w = x2 - x1
        # second box = the first shifted right by d: IoU = (w - d) / (w + d) = thr (+- rounding)  ->  d = w (1 - thr) / (1 + thr)
        d = np.float32(w * (np.float32(1) - thr) / (np.float32(1) + thr)) * np.float32(1 + (rng_.integers(-3, 4)) * 2.0 ** -22)
        nt_b[q + 1] = [x1 + d, y1, x2 + d, y2]
        nt_s[q + 1] = nt_s[q] - np.float32(1e-3)
    cases.append((nt_b, nt_s, 0.01, 0.6))
    for k, (boxes, sims, conf, iou) in enumerate(cases):
        res[f"boxes_{k}"] = boxes; res[f"sims_{k}"] = sims
        res[f"conf_{k}"] = np.float32(conf); res[f"iou_{k}"] = np.float32(iou)
        for route, tag in (("per_class", "out"), ("coordinate_offset", "off_out")):
            _NMS_ROUTE[0] = route
            pp = RefPostProcess(confidence_threshold=conf, iou_threshold=iou)
            ob, oc, os_ = pp(torch.from_numpy(boxes)[None].clone(), torch.from_numpy(sims)[None].clone())
            res[f"{tag}_boxes_{k}"] = ob.numpy(); res[f"{tag}_classes_{k}"] = oc.numpy(); res[f"{tag}_scores_{k}"] = os_.numpy()
            print("f6 case", k, route, "kept", ob.shape[1], "of", boxes.shape[0])
    _NMS_ROUTE[0] = "per_class"
    res["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "f6_postprocess.npz"), **res)


def f7():
    """Input pipeline (ref src/dataset.py:69-71): PIL bicubic + the HF OwlViTImageProcessor on u8 RGB images.
    Big cases are stored sub-sampled (every `stride`-th row/column) plus a checksum of the full resized image."""
    from PIL import Image
    from transformers import OwlViTImageProcessor
    rng_ = np.random.default_rng(2024)
    res = {}
    cases = [(37, 53, 96, 1), (200, 333, 96, 1), (97, 131, 96, 1), (5, 7, 96, 1), (96, 96, 96, 1),
             (480, 640, 768, 13), (1000, 1500, 768, 13), (427, 640, 840, 17)]
    for k, (H, W, S, st) in enumerate(cases):
        # smooth-ish content + noise so that both interpolation and clamping paths are exercised
        yy, xx = np.mgrid[0:H, 0:W]
        base = 127 + 120 * np.sin(xx / 9.0 + k)[..., None] * np.cos(yy / 7.0)[..., None] * np.ones(3)
        img = np.clip(base + rng_.normal(0, 40, (H, W, 3)), 0, 255).astype(np.uint8)
        if k % 2 == 0:
            img = rng_.integers(0, 256, (H, W, 3), dtype=np.uint8)
        seeded = H * W > 50000
        if seeded:      # big inputs are regenerated from the repo's counter-based RNG instead of being stored
            from owl_vit_object_detection_amd import rng as crng
            img = crng.randint(77, f"f7/{k}", H * W * 3, 256).reshape(H, W, 3).astype(np.uint8)
        pil = Image.fromarray(img)
        resized = np.asarray(pil.resize((S, S), resample=Image.BICUBIC))
        ip = OwlViTImageProcessor(size={"height": S, "width": S})
        pv = ip(images=pil, return_tensors="pt")["pixel_values"][0].numpy()
        assert pv.shape == (3, S, S)
        if seeded:
            res[f"shape_{k}"] = np.array([H, W], np.int64)
        else:
            res[f"img_{k}"] = img
        res[f"size_{k}"] = np.int64(S); res[f"stride_{k}"] = np.int64(st)
        res[f"resized_{k}"] = resized[::st, ::st].copy(); res[f"resized_sum_{k}"] = np.int64(resized.astype(np.int64).sum())
        res[f"pixel_values_{k}"] = pv[:, ::st, ::st].copy()
        print("f7 case", k, (H, W, S), "mean", float(pv.mean()))
    res["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "f7_preprocess.npz"), **res)


def synth_prompt_ids(tc, n_prompts, seed=5):
    """CLIP-processor-shaped token ids: BOS, 1..8 word tokens, EOS (= highest id), zero padding to max_pos."""
    from owl_vit_object_detection_amd import rng as crng
    S = tc.max_pos
    ids = np.zeros((n_prompts, S), np.int64)
    lens = 1 + crng.randint(seed, "prompt/len", n_prompts, min(8, S - 2))
    words = crng.randint(seed, "prompt/words", n_prompts * S, tc.vocab - 3).reshape(n_prompts, S) + 1
    for n in range(n_prompts):
        ids[n, 0] = tc.vocab - 2
        ids[n, 1:1 + lens[n]] = words[n, :lens[n]]
        ids[n, 1 + lens[n]] = tc.vocab - 1
    return ids


def f8():
    """Query-bank init: HF OwlViTForObjectDetection(...).text_embeds exactly as ref src/models.py:161-169 calls it
    (input_ids + the processor's padding attention_mask + a dummy image), text weights from weights.make_text_weights."""
    from owl_vit_object_detection_amd.config import get_text_config
    res = {}
    for cname, n_prompts in (("tiny", 12), ("owlvit-base-patch16", 30)):
        tc = get_text_config(cname)
        vcfg = get_config("tiny")
        hf_cfg = OwlViTConfig(
            vision_config=dict(hidden_size=vcfg.hidden, intermediate_size=vcfg.mlp, num_hidden_layers=2,
                               num_attention_heads=vcfg.heads, image_size=vcfg.image_size, patch_size=vcfg.patch_size),
            text_config=dict(hidden_size=tc.width, intermediate_size=tc.mlp, num_hidden_layers=tc.layers,
                             num_attention_heads=tc.heads, vocab_size=tc.vocab, max_position_embeddings=tc.max_pos,
                             hidden_act="quick_gelu", layer_norm_eps=tc.ln_eps, bos_token_id=tc.vocab - 2,
                             eos_token_id=tc.vocab - 1, pad_token_id=0),
            projection_dim=tc.proj_dim,
        )
        hf_cfg._attn_implementation = "eager"
        hf_cfg.text_config._attn_implementation = "eager"
        hf_cfg.vision_config._attn_implementation = "eager"
        hf = OwlViTForObjectDetection(hf_cfg).eval()
        tw = weights.make_text_weights(tc)
        sd = hf.state_dict()
        for name, arr in tw.items():
            key = "owlvit." + name
            assert key in sd and tuple(sd[key].shape) == arr.shape, (key, arr.shape)
            sd[key] = torch.from_numpy(arr)
        hf.load_state_dict(sd)
        ids = synth_prompt_ids(tc, n_prompts)
        mask = (np.arange(tc.max_pos)[None, :] <= ids.argmax(1)[:, None]).astype(np.int64)
        with torch.no_grad():
            out = hf(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                     pixel_values=torch.zeros(1, 3, vcfg.image_size, vcfg.image_size))
        te = out.text_embeds
        assert te.shape == (1, n_prompts, tc.proj_dim), te.shape
        res[cname + "/input_ids"] = ids
        res[cname + "/text_embeds"] = te[0].numpy()
        print("f8", cname, te.shape, "row norms", float(te[0].norm(dim=-1).mean()), "cos(0,1)", float((te[0, 0] * te[0, 1]).sum()))
    np.savez_compressed(os.path.join(HERE, "f8_text_embeds.npz"), **res)


def lsap():
    """Known-answer vectors from scipy (the reference's solver) for the C restatement."""
    from scipy.optimize import linear_sum_assignment
    from owl_vit_object_detection_amd import rng
    res = {}
    shapes = [(1, 1), (5, 5), (2304, 7), (7, 2304), (300, 40), (64, 64), (10, 3), (3, 10), (576, 16)]
    for k, (nr, nc) in enumerate(shapes):
        c = rng.uniform(11, f"lsap/{k}", nr * nc).reshape(nr, nc)
        i, j = linear_sum_assignment(c)
        res[f"c{k}/cost"] = c; res[f"c{k}/row"] = i; res[f"c{k}/col"] = j
    for k, (nr, nc) in enumerate([(6, 6), (40, 8), (8, 40), (100, 10)]):
        c = rng.randint(11, f"lsapint/{k}", nr * nc, 4).reshape(nr, nc).astype(np.float64)
        i, j = linear_sum_assignment(c)
        res[f"t{k}/cost"] = c; res[f"t{k}/row"] = i; res[f"t{k}/col"] = j
    np.savez_compressed(os.path.join(HERE, "lsap_cases.npz"), **res)
    print("lsap cases:", len(res) // 3)


def f9():
    """F9 -- the input contract: reference `coco_to_model_input` / `model_output_to_image`
    (src/train_util.py:4-24 -> BoxUtil.box_convert / scale_bounding_box, src/util.py:83-93,123-129) run on DataLoader-shaped
    inputs: boxes [1,n,4] f32 absolute COCO xywh, metadata = default_collate of {"width": int, "height": int}.
    `src/util.py` imports tensorboard / torchvision.io / torchvision.utils at module level (absent here): those are stubbed with
    empty placeholders (never called on this path).  `torchvision.ops.box_convert` IS called; torchvision is a third-party
    dependency absent from /root/reference, so its published xywh->xyxy rule (x, y, x+w, y+h via unbind/stack,
    torchvision/ops/_box_convert.py `_box_xywh_to_xyxy`) is what the stub restates -- that one line is "parity unpinned" for
    torchvision; the division by (width, height), its in-place behaviour, dtypes and shapes are the reference's own code."""
    tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    tvio = types.ModuleType("torchvision.io"); tvio.read_image = None
    tvu = types.ModuleType("torchvision.utils"); tvu.draw_bounding_boxes = None
    sys.modules["torchvision.io"] = tvio; sys.modules["torchvision.utils"] = tvu
    tv.io = tvio; tv.utils = tvu

    def _box_convert(boxes, in_fmt, out_fmt):
        assert (in_fmt, out_fmt) == ("xywh", "xyxy")
        x, y, w, h = boxes.unbind(-1)
        return torch.stack([x, y, x + w, y + h], dim=-1)
    tv_ops.box_convert = _box_convert
    from src.train_util import coco_to_model_input as ref_c2m, model_output_to_image as ref_m2i
    from torch.utils.data import default_collate
    from owl_vit_object_detection_amd import rng
    res = {}
    sizes = [(640, 480), (500, 375), (333, 500), (1, 1), (4032, 3024), (427, 640)]
    for k, (w, h) in enumerate(sizes):
        n = 1 + k * 3
        x0 = rng.uniform(17, f"f9/{k}", n, 0) * w * 0.7; y0 = rng.uniform(17, f"f9/{k}", n, 1) * h * 0.7
        bw = 1.0 + rng.uniform(17, f"f9/{k}", n, 2) * w * 0.3; bh = 1.0 + rng.uniform(17, f"f9/{k}", n, 3) * h * 0.3
        xywh = np.round(np.stack([x0, y0, bw, bh], 1), 2).astype(np.float32)          # COCO annotations carry 2 decimals
        boxes = default_collate([torch.tensor(xywh.tolist())])                        # [1, n, 4] f32, as the DataLoader yields
        meta = default_collate([{"width": w, "height": h}])                           # {"width": tensor([w]), "height": tensor([h])}
        inp = boxes.clone()
        out = ref_c2m(inp, meta)
        assert torch.equal(inp, boxes)                                                # the reference does not touch its input here
        back = ref_m2i(out.clone(), meta)
        res[f"c{k}/xywh"] = boxes.numpy(); res[f"c{k}/wh"] = np.array([w, h], np.int64)
        res[f"c{k}/xyxy_norm"] = out.numpy(); res[f"c{k}/back"] = back.numpy()
    np.savez_compressed(os.path.join(HERE, "f9_input_contract.npz"), **res)
    print("f9 cases:", len(sizes))


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f3", "f5", "lsap", "f2", "f4"]
    for w in which:
        globals()[w]()
