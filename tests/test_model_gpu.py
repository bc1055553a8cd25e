"""End-to-end parity on a real MI355X: HIP path (through the reference call surface) vs the CPU oracle
on the same seeded weights/inputs, and vs the committed reference-generated fixtures.
Tolerance: the north star's bf16 bar -- outputs within 1e-2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import owl_oracle as O  # noqa: E402  (checker only)
from owl_vit_object_detection_amd import synth, weights  # noqa: E402
from owl_vit_object_detection_amd.config import get_config  # noqa: E402
from owl_vit_object_detection_amd.models import OwlViT  # noqa: E402

DEV = "cuda"
# north-star bf16 bar: 1e-2.  Asserted at ~2x the error measured on the final build (tools/errstudy.py: full-size B/16 boxes 2.0e-3 /
# sims 9.1e-4, L/14 1.8e-3 / 7.3e-4) so that a regression which doubles the forward error fails.
# Measured on the round-2 build (gpurun_out/r2_t1.log): full size boxes <= 2.4e-3 / sims <= 9.1e-4 (B/16, all 8 images of configs[1]),
# 1.9e-3 / 6.7e-4 (L/14), 1.7e-3 / 7.2e-4 (B/32); parity-test configs (tiny / small: 2 heads, D = 128 / 256, where one bf16 ulp of a
# feature is a larger share of a cosine) boxes <= 1.7e-3 / sims <= 1.8e-3.
TOL_BOXES, TOL_SIMS, TOL_SIMS_SMALLCFG = 4e-3, 2e-3, 3.5e-3
# End-to-end gradient bands of the full-size fixtures: per tensor |norm ratio - 1| and the cosine on a 4096-element strided
# sample of the WHOLE tensor (fixture key gradsample/), asserted at ~2x what the round-4 build measures (profiles/r04_parity_bands.md; the numbers
# are printed by every run).  Default band for every tensor; the named ones need more and say why.  The strict all-element check is the
# backward-chain test (measured <= 9.7e-3 rel-L2 at HF-init weights).
# Measured (round 4):
#   F4 (L/14): every tensor within 7.4e-3 / cosine >= 0.99606.
#   F2 (B/16): tensors fed by the CLASS loss only (queries, class_predictor.*) within 1.4e-3 / 0.99998.  Matched row 450 of this fixture sits 2e-4 from
#   its target in x0 (reference) -- the bf16 forward puts it at -4e-4, sign(p - t) of the L1 term flips, and that row carries the largest box
#   gradient of the image (|d_box| 2.0 against 0.1-0.5 for the other twelve: tests/diag_box_grad.py).  Every tensor downstream of the BOX loss inherits
#   it: backbone / LayerNorm tensors 2.6e-2 ... 7.7e-2 / 0.942 ... 0.967, box_head.* unrelated to the reference (skipped, as before).  The box path is
#   pinned instead by test_loss_backward_at_the_operating_point (HIP d_boxes / d_sims == the oracle's autograd AT THE HIP OUTPUTS, 1e-7) and by the
#   backward-chain test (same upstream into both backwards, <= 9.5e-3 on every tensor, batch 8 of B/16 included).
GRAD_BANDS_DEFAULT = (1.5e-2, 0.992)                # (|norm ratio - 1|, min cosine on the 4096-element sample)
_CLASS_ONLY = ("queries", "class_predictor.dense0.weight", "class_predictor.dense0.bias")
GRAD_BANDS_F2 = {n: (3e-3, 0.9999) for n in _CLASS_ONLY}
GRAD_BANDS_F2_BOX_FED = (0.155, 0.88)               # 2x (7.7e-2, 1 - 0.942): see above
GRAD_BANDS_F4 = {}
# losses: F2 <= 8.0e-4 (loss_ce 3.1e-4), F4 <= 5.0e-4 relative; parity-test configs (36-144 patches: one bf16 ulp of a sim is a larger share of a
# loss): tiny B=1 loss_ce 9.75e-3, tiny B=3 4.4e-3, tiny-l14 3.2e-3, small 7e-4
LOSS_REL_SMALLCFG = 2e-2
LOSS_REL = 2e-3


def _check_grads_vs_fixture(g, grads, tag, bands, skip=lambda n: False):
    """Per-tensor comparison of the end-to-end gradients with a reference fixture: norm ratio (gradnorm/) + cosine on the 4096-element strided
    sample (gradsample/).  Tensors whose reference norm is below 1e-5 (k_proj.bias: the true gradient is zero) must be ~zero."""
    worst_norm, worst_cos, lines, bad = 0.0, 1.0, [], []
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        if ref_norm < 1e-5:                  # (k_proj.bias: the true gradient is zero -- judged against the image's largest gradient, 1e-3 at HF-init scales)
            assert float(gr.double().norm()) < max(1e-3, 2e-4 * max(float(g["gradnorm/" + m]) for m in grads)), n
            continue
        ratio = float(gr.double().norm()) / ref_norm
        ref_s = torch.from_numpy(g["gradsample/" + n]).double()
        ours_s = _sample(gr).double()
        cos = float((ours_s * ref_s).sum() / (ours_s.norm() * ref_s.norm() + 1e-30))
        skipped = skip(n)
        lines.append(f"     {n:58s} |ref|={ref_norm:.3e} |norm ratio - 1| {abs(ratio - 1):.3e} cos(4096 sample) {cos:.5f}" + ("  (skipped: near-tie)" if skipped else ""))
        if skipped:
            continue
        bn, bc = bands.get(n, bands.get("*", GRAD_BANDS_DEFAULT))
        worst_norm = max(worst_norm, abs(ratio - 1.0)); worst_cos = min(worst_cos, cos)
        if abs(ratio - 1.0) > bn or cos < bc:
            bad.append((n, abs(ratio - 1.0), cos, bn, bc))
    print("\n".join(lines))
    print(f"{tag} end-to-end gradients: worst |norm ratio - 1| = {worst_norm:.3e}, worst 4096-sample cosine = {worst_cos:.5f}")
    assert not bad, bad


def _tol_sims(cname):
    return TOL_SIMS if cname.startswith("owlvit") else TOL_SIMS_SMALLCFG


def _maxerr(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


@pytest.mark.parametrize("cname,B", [("tiny", 1), ("tiny", 3), ("small", 2), ("tiny-l14", 2)])
def test_forward_matches_oracle(cname, B):
    cfg = get_config(cname)
    Wnp = weights.make_weights(cfg)
    model = OwlViT(cfg, Wnp, DEV)
    img = synth.make_images(cfg, B)
    with torch.no_grad():
        pb, n1, ps, n2 = model(torch.from_numpy(img).to(DEV))
    assert n1 is None and n2 is None and pb.shape == (B, cfg.patches, 4) and ps.shape == (B, cfg.patches, cfg.n_classes)
    assert pb.dtype == torch.float32 and ps.dtype == torch.float32
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    rb, rs = O.model_forward(cfg, w, torch.from_numpy(img))
    eb, es = _maxerr(pb, rb), _maxerr(ps, rs)
    print(f"{cname} B={B}: max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
    assert eb < TOL_BOXES and es < _tol_sims(cname), (eb, es)


def test_forward_matches_reference_fixture_f1(golden_dir):
    cfg = get_config("tiny")
    g = np.load(os.path.join(golden_dir, "f1_tiny.npz"))
    model = OwlViT(cfg, weights.make_weights(cfg), DEV)
    img = torch.from_numpy(synth.make_images(cfg, 1)).to(DEV)
    with torch.no_grad():
        pb, _, ps, _ = model(img)
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    print(f"tiny vs reference fixture F1: max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
    assert eb < TOL_BOXES and es < TOL_SIMS_SMALLCFG, (eb, es)


def test_forward_b16_matches_reference_fixture_f2(golden_dir):
    """BASELINE configs[1] shape family: owlvit-base-patch16 768x768 forward vs the reference's CPU logits."""
    cfg = get_config("owlvit-base-patch16")
    g = np.load(os.path.join(golden_dir, "f2_b16.npz"))
    model = OwlViT(cfg, weights.make_weights(cfg), DEV)
    img = torch.from_numpy(synth.make_images(cfg, 1)).to(DEV)
    with torch.no_grad():
        pb, _, ps, _ = model(img)
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    print(f"B/16 vs reference fixture: max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
    assert eb < TOL_BOXES and es < TOL_SIMS, (eb, es)
    # batch invariance: image 0 inside a batch of 8 gives the same outputs
    imgs = torch.from_numpy(synth.make_images(cfg, 8)).to(DEV)
    with torch.no_grad():
        pb8, _, ps8, _ = model(imgs)
    assert _maxerr(pb8[0], pb[0]) < 1e-6 and _maxerr(ps8[0], ps[0]) < 1e-6


def test_forward_b32_default_arch_matches_oracle():
    """What the reference actually loads (src/models.py:152: google/owlvit-base-patch32, 24 x 24 patches of 32 px, T = 577):
    `load_model`'s default arch, through the reference call surface, against the CPU oracle on the same weights."""
    from owl_vit_object_detection_amd.models import load_model
    labelmap = {str(i): i for i in range(10)}
    model = load_model(labelmap, DEV).eval()
    cfg = model.cfg
    assert cfg.name == "owlvit-base-patch32" and cfg.patches == 576 and cfg.tokens == 577
    img = synth.make_images(cfg, 2)
    with torch.no_grad():
        pb, n1, ps, n2 = model(torch.from_numpy(img).to(DEV))
    assert n1 is None and n2 is None and pb.shape == (2, 576, 4) and ps.shape == (2, 576, 10)
    w = {k: torch.from_numpy(v) for k, v in weights.make_weights(cfg).items()}
    rb, rs = O.model_forward(cfg, w, torch.from_numpy(img))
    eb, es = _maxerr(pb, rb), _maxerr(ps, rs)
    print(f"B/32: max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
    assert eb < TOL_BOXES and es < TOL_SIMS, (eb, es)


# ---------------------------------------------------------------------------------------------------
# full train step: forward + matcher/loss + backward through the reference call surface
# ---------------------------------------------------------------------------------------------------
from owl_vit_object_detection_amd.losses import PushPullLoss  # noqa: E402

LOSS_KEYS = ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")


def _step_hip(cfg, Wnp, img, labels, boxes, scales):
    model = OwlViT(cfg, Wnp, DEV)
    crit = PushPullLoss(cfg.n_classes, scales)
    pb, _, ps, _ = model(torch.from_numpy(img).to(DEV))
    losses = crit(ps, [torch.from_numpy(l).to(DEV) for l in labels], pb, [torch.from_numpy(b).to(DEV) for b in boxes])
    loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]   # ref main.py:84-89
    loss.backward()
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}
    return model, crit, {k: float(v) for k, v in losses.items()}, grads, pb.detach(), ps.detach()


def _grad_report(grads, ref, tag):
    """rel-L2 per tensor, measured against max(|ref|, 1e-3 * largest |ref|): tensors whose true gradient
    is ~0 (k_proj.bias: softmax is invariant to a key bias) are judged on an absolute scale."""
    floor = 1e-3 * max(float(r.float().norm()) for r in ref.values())
    worst, worst_cos = 0.0, 1.0
    lines = []
    for n, r in ref.items():
        g = grads[n]
        r = r.float()
        rel = float((g - r).norm() / max(float(r.norm()), floor))
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-20)) if float(r.norm()) > floor else 1.0
        lines.append(f"  {n:58s} rel_l2={rel:.3e} cos={cos:.5f} |ref|={float(r.norm()):.3e}")
        worst = max(worst, rel)
        worst_cos = min(worst_cos, cos)
    print(f"[{tag}] worst rel-L2 grad error {worst:.3e}, worst cos {worst_cos:.5f}\n" + "\n".join(lines))
    return worst, worst_cos


@pytest.mark.parametrize("cname,B", [("tiny", 1), ("tiny", 3), ("small", 2), ("tiny-l14", 2), ("owlvit-base-patch32", 2), ("owlvit-base-patch16", 1), ("owlvit-large-patch14", 1),
                                     ("owlvit-base-patch16", 8)])       # batch 8 of B/16: the batched backward tied to the oracle directly, not via batch-1 self-consistency
def test_backward_chain_matches_oracle_given_same_upstream(cname, B):
    """Backward kernels in isolation: identical upstream (d_boxes, d_sims) into the HIP backward and into
    the oracle's autograd -- removes the loss's 1/|sim| amplification of bf16 forward noise."""
    cfg = get_config(cname)
    torch.set_num_threads(min(32, os.cpu_count() or 1))     # the oracle's CPU backward at full B/16 size
    Wnp = weights.make_weights(cfg)
    img = synth.make_images(cfg, B)
    g = torch.Generator().manual_seed(5)
    d_boxes = torch.randn(B, cfg.patches, 4, generator=g) * 0.1
    d_sims = torch.randn(B, cfg.patches, cfg.n_classes, generator=g) * 0.1
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    names = O.trainable_names(w)
    ww = {n: (t.clone().requires_grad_(True) if n in names else t) for n, t in w.items()}
    taps = {}
    rb, rs = O.model_forward(cfg, ww, torch.from_numpy(img), taps)
    # MaxPool1d(3) routes each class gradient to ONE of three prompts; where the top two prompts are within
    # bf16 forward noise of each other the routing is a coin flip, so those (row, class) pairs get no upstream
    with torch.no_grad():
        e = torch.nn.functional.linear(taps["feats"], w["class_predictor.dense0.weight"], w["class_predictor.dense0.bias"])
        e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
        q = w["queries"] / torch.linalg.norm(w["queries"], dim=-1, keepdim=True) + 1e-6
        top2 = (e @ q.transpose(1, 2)).view(B, cfg.patches, cfg.n_classes, 3).topk(2, dim=-1).values
        d_sims = d_sims * ((top2[..., 0] - top2[..., 1]) > 0.02).float()
    model = OwlViT(cfg, Wnp, DEV)
    pb, _, ps, _ = model(torch.from_numpy(img).to(DEV))
    torch.autograd.backward([pb, ps], [d_boxes.to(DEV), d_sims.to(DEV)])
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}
    torch.autograd.backward([rb, rs], [d_boxes, d_sims])
    gref = {n: ww[n].grad for n in names}
    worst, worst_cos = _grad_report(grads, gref, f"backward-only {cname} B={B}")
    assert worst < 2e-2 and worst_cos > 0.9995          # measured <= 9.7e-3 / >= 0.99996 on every config


@pytest.mark.parametrize("cname,B", [("tiny", 1), ("tiny", 3), ("small", 2)])
def test_train_step_matches_oracle(cname, B):
    cfg = get_config(cname)
    Wnp = weights.make_weights(cfg)
    img = synth.make_images(cfg, B)
    labels, boxes = synth.make_targets(cfg, B, max_boxes=6)
    scales = synth.class_scales(cfg, labels)
    model, crit, lg, grads, pb, ps = _step_hip(cfg, Wnp, img, labels, boxes, scales)
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    (rb, rs), lo, gref = O.train_step(cfg, w, torch.from_numpy(img), [torch.from_numpy(l) for l in labels],
                                      [torch.from_numpy(b) for b in boxes], torch.from_numpy(scales))
    eb, es = _maxerr(pb, rb), _maxerr(ps, rs)
    print(f"train step {cname} B={B}: max|d boxes|={eb:.3e} max|d sims|={es:.3e}")
    assert eb < TOL_BOXES and es < _tol_sims(cname), (eb, es)
    # matched loss within the bf16 bar (relative for the large class terms)
    for k in LOSS_KEYS:
        print(f"   {k}: {lg[k]:.6f} oracle {float(lo[k]):.6f} rel {abs(lg[k] - float(lo[k])) / abs(float(lo[k])):.2e}")
        assert abs(lg[k] - float(lo[k])) <= LOSS_REL_SMALLCFG * abs(float(lo[k])), (k, lg[k], float(lo[k]))
    assert set(grads) == set(gref) and len(grads) == 29
    # end-to-end gradients are a sanity check only: the class terms' -w/|sim| slope and the max-over-prompts
    # routing amplify the ~1e-3 bf16 forward deviation (the strict kernel check is the backward-only test)
    worst, worst_cos = _grad_report(grads, gref, f"{cname} B={B}")
    assert worst_cos > 0.99


@pytest.mark.parametrize("cname", ["tiny", "tiny-l14"])
def test_train_step_matches_reference_fixture_f1(golden_dir, cname):
    """tiny-l14 has two frozen layers ABOVE the trainable layer 11 -> exercises the dX-only backward (L/14's case)."""
    cfg = get_config(cname)
    g = np.load(os.path.join(golden_dir, f"f1_{cname}.npz"))
    img = synth.make_images(cfg, 1)
    labels, boxes = synth.make_targets(cfg, 1, max_boxes=6)
    model, crit, lg, grads, pb, ps = _step_hip(cfg, weights.make_weights(cfg), img, labels, boxes, g["scales"])
    assert np.array_equal(crit.last["target_classes"][0].cpu().numpy(), g["target_classes"])
    for k in LOSS_KEYS:
        print(f"   {k}: {lg[k]:.6f} ref {float(g[k]):.6f} rel {abs(lg[k] - float(g[k])) / abs(float(g[k])):.2e}")
        assert abs(lg[k] - float(g[k])) <= LOSS_REL_SMALLCFG * abs(float(g[k])), k
    ref = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad/")}
    worst, worst_cos = _grad_report(grads, ref, f"{cname} vs reference fixture")
    assert worst_cos > 0.99


def _class_loss_bound(cfg, g, es):
    """First-order bound on |d loss_ce|, |d loss_bg| for a forward deviation of max|d sims| = es:
    |dL| <= es * ||dL/d sims||_1 (Hoelder), with the gradient taken at the REFERENCE outputs stored in the fixture.
    The class terms carry a -w/|sim| slope, so a ~1e-3 bf16 deviation of a sim near 0 moves them by a few percent;
    the bound says exactly how much is explainable by the measured forward error (x2 for second-order slack)."""
    sims = torch.from_numpy(g["pred_sims"][0]).clone().requires_grad_(True)
    tc = torch.from_numpy(g["target_classes"]).long()
    ce, bgl = O.class_loss(sims, tc, cfg.n_classes, torch.from_numpy(g["scales"]).float())
    out = {}
    for k, v in (("loss_ce", ce), ("loss_bg", bgl)):
        (gr,) = torch.autograd.grad(v, sims, retain_graph=True)
        out[k] = 2.0 * es * float(gr.abs().sum())
    return out


def _box_loss_first_order(g, boxes, pb):
    """The two box terms respond to the forward deviation of the MATCHED rows only: first-order prediction sum(dL/d(box) * (box_hip - box_ref)) at the reference's
    outputs, and its magnitude sum|dL/d(box) * delta| (the scale of what second order may add).  The margin fixtures' targets hug their predictions (edges
    6e-3 ... 1.5e-2 apart), so a 1e-3 deviation is 10 % of an L1 distance: compared like with like, the loss kernel is held to LOSS_REL + 15 % of that scale."""
    pi, ti = torch.from_numpy(g["pred_idx"]).long(), torch.from_numpy(g["tgt_idx"]).long()
    src = torch.from_numpy(g["pred_boxes"][0])[pi].clone().requires_grad_(True)
    dst = torch.from_numpy(boxes[0]).float()[ti]
    delta = pb[0].cpu().float()[pi] - src.detach()
    n = src.shape[0]
    l1 = (src - dst).abs().sum() / n
    giou = (1 - torch.diag(O.generalized_box_iou(src, dst))).sum() / n
    out = {}
    for k, v in (("loss_bbox", l1), ("loss_giou", giou)):
        (gr,) = torch.autograd.grad(v, src, retain_graph=True)
        out[k] = (float((gr * delta).sum()), float((gr * delta).abs().sum()))
    return out


def _reference_losses_for_decisions(cfg, g, labels, boxes, crit, eb, es):
    """The step contains DISCRETE decisions -- the Hungarian assignment (argmin of a cost) and the label spreading
    (IoU > 0.85) -- that legitimately land on the other side of a near-tie when the bf16 forward deviates by ~1e-3
    from the fp32 reference (L/14 fixture: one assignment swap = 2 of 3600 label rows).  This helper (1) checks that
    every decision that differs from the fixture IS a near-tie in the reference's own numbers (assignment: cost gap on
    the reference's cost matrix below twice the per-entry cost uncertainty implied by the measured forward error --
    L1 term 4*eb, class term es, GIoU term ~8*eb; spreading: |IoU - 0.85| < 2e-2 on the reference's boxes) and (2) returns
    the four losses of the REFERENCE outputs scored with this run's decisions, so losses are compared like with like."""
    n_cls = cfg.n_classes
    ref_sims, ref_boxes = torch.from_numpy(g["pred_sims"][0]), torch.from_numpy(g["pred_boxes"][0])
    lab, tgt = torch.from_numpy(labels[0]).long(), torch.from_numpy(boxes[0]).float()
    n = lab.shape[0]
    tc_ours = crit.last["target_classes"][0].cpu().long()
    pi_ours, ti_ours = crit.last["pred_idx"][0, :n].cpu().long(), crit.last["tgt_idx"][0, :n].cpu().long()
    pi_ref, ti_ref = torch.from_numpy(g["pred_idx"]).long(), torch.from_numpy(g["tgt_idx"]).long()
    C, _, _, _ = O.match_one(ref_sims, ref_boxes, lab, tgt, n_cls)
    C = torch.as_tensor(C)
    row_ours = torch.empty(n, dtype=torch.long); row_ours[ti_ours] = pi_ours
    row_ref = torch.empty(n, dtype=torch.long); row_ref[ti_ref] = pi_ref
    swaps = torch.nonzero(row_ours != row_ref).flatten()
    for t in swaps.tolist():
        gap = abs(float(C[row_ours[t], t]) - float(C[row_ref[t], t]))
        assert gap < 2.0 * (12.0 * eb + es), (t, gap, eb, es)    # assignment differs only across a cost near-tie
    tc_ref = torch.from_numpy(g["target_classes"]).long()
    diff = torch.nonzero(tc_ours != tc_ref).flatten()
    assert diff.numel() <= max(2, int(0.002 * tc_ref.numel())), diff.numel()
    explained = set(row_ours[swaps].tolist()) | set(row_ref[swaps].tolist())
    rest = [r for r in diff.tolist() if r not in explained]
    if rest:
        pos = torch.nonzero((tc_ref != n_cls) | (tc_ours != n_cls)).flatten()
        iou = O.box_iou(ref_boxes[rest], ref_boxes[pos])[0]
        assert float((iou - 0.85).abs().min(dim=1).values.max()) < 2e-2
    ce, bgl = O.class_loss(ref_sims, tc_ours, n_cls, torch.from_numpy(g["scales"]).float())
    src, dst = ref_boxes[pi_ours], tgt[ti_ours]
    l1 = float((src - dst).abs().sum() / n)
    giou = float((1 - torch.diag(O.generalized_box_iou(src, dst))).sum() / n)
    return {"loss_ce": float(ce), "loss_bg": float(bgl), "loss_bbox": l1, "loss_giou": giou}, int(swaps.numel()), int(diff.numel())


def _near_tie(g, boxes):
    """sign(pred - tgt) in the L1 term and the min/max selections in GIoU are discontinuous: a matched coordinate
    within bf16-forward noise of its target legitimately flips them relative to the fp32 reference, and that single
    row then dominates the box-head gradient norms."""
    pi, ti = g["pred_idx"], g["tgt_idx"]
    return bool((np.abs(g["pred_boxes"][0][pi] - boxes[0][ti]) < 3e-3).any())


def test_train_step_b16_matches_reference_fixture_f2(golden_dir):
    """BASELINE configs[2] shape family at batch 1: full train step of owlvit-base-patch16 vs the
    reference's CPU run (losses, per-tensor gradient norms and leading elements)."""
    cfg = get_config("owlvit-base-patch16")
    g = np.load(os.path.join(golden_dir, "f2_b16.npz"))
    img = synth.make_images(cfg, 1)
    labels, boxes = synth.make_targets(cfg, 1, max_boxes=16)
    model, crit, lg, grads, pb, ps = _step_hip(cfg, weights.make_weights(cfg), img, labels, boxes, g["scales"])
    same = float((crit.last["target_classes"][0].cpu() == torch.from_numpy(g["target_classes"])).float().mean())
    print("B/16 losses", lg, "ref", {k: float(g[k]) for k in LOSS_KEYS}, "target_classes agreement", same)
    assert same == 1.0
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    assert eb < TOL_BOXES and es < TOL_SIMS, (eb, es)
    for k in LOSS_KEYS:
        print(f"   {k}: {lg[k]:.6f} ref {float(g[k]):.6f} rel {abs(lg[k] - float(g[k])) / abs(float(g[k])):.2e}")
        assert abs(lg[k] - float(g[k])) <= LOSS_REL * abs(float(g[k])), (k, lg[k], float(g[k]))
    near_tie = _near_tie(g, boxes)
    assert near_tie                      # (the fixture's row 450; if the fixture changes, the box-fed band below goes back to the default)
    _check_grads_vs_fixture(g, grads, "B/16 F2", dict(GRAD_BANDS_F2, **{"*": GRAD_BANDS_F2_BOX_FED}), skip=lambda n: near_tie and n.startswith("box_head"))


# ---------------------------------------------------------------------------------------------------
# F2b / F4b / F10b (+ F2c): the end-to-end gradient pin against the REFERENCE at full size, on fixtures whose decisions and loss kinks have margin
# (VERDICT r04 #1).  F2 / F4 / F10 above all have a matched coordinate within 3e-3 of its target, so box_head.* is skipped there and every box-fed
# tensor takes the loose band; here nothing is skipped and every one of the 29 tensors is held to GRAD_BANDS_DEFAULT.  The generator
# (tests/golden/make_golden.py: anchored_targets / decision_margins) verifies on the reference's own outputs: matched coordinates >= 5e-3 from their
# targets, GIoU selections >= 5e-3, runner-up assignment >= 2.5e-2 (= 10x the forward error) more expensive, no |IoU - 0.85| < 2e-2 in the spreading
# scan, |sim| >= 2e-2 on every positive row; the margins travel in the fixture (margin/*) and the targets too (tgt_labels / tgt_boxes).
# F2c is what a pure SEED search over synth.make_targets reaches (criteria 1, 2, 4 hold; the runner-up gap does not reach the bar -- at HF-init weights every
# prediction inside a seeded target costs the same): asserted the same way when its decisions agree, like with like otherwise.
# ---------------------------------------------------------------------------------------------------
MARGIN_FIXTURES = [("owlvit-base-patch16", "f2b_b16_margins", "init"), ("owlvit-large-patch14", "f4b_l14_margins", "init"),
                   ("owlvit-base-patch16", "f10b_b16_trained_margins", "trained_like"), ("owlvit-base-patch16", "f2c_b16_seed_search", "init")]
# Bands (|norm ratio - 1|, min cosine on the 4096-element sample), ~2x what the round-5 build measures (profiles/r05_parity_bands.md; every run prints them).
# The end-to-end deviation is ONE DRAW of the bf16 forward's rounding: the same build with another summation order in the patch embedding alone (L/14, the
# im2row-free K order of this round) moved F4b's box-head tensors from <= 9.7e-3 to 1.5-1.8e-2 and dense2.bias from 1.64e-2 to 9e-3 -- the band has to hold the
# spread of the draws, not one of them.
#   F2b: every tensor within 1.41e-2 / cosine >= 0.99991.   F4b: 0.97e-2 ... 1.80e-2 / >= 0.99944 over the two draws; box_head.dense2.bias (FOUR numbers = sum over the
#   12 matched rows of d_box * sigmoid': the GIoU slope moves by eb / box size = 2e-3 / 0.03 per row and a 4-element sum does not average it) up to 1.64e-2.
#   F2c (seed search: |sim| margin only 1.2e-2, so the class term's -w / |sim| slope moves 8 % on one row): every backbone tensor 1.4-1.6e-2 in norm, cosine >= 0.99987.
#   F10b (trained-like weights): worst 3.74e-2 (layer 11 k_proj.weight) / 0.99824; class-only tensors 1.5e-3 / 1.00000, box head 0.9-2.4e-2.  The backward-chain test on
#   the same weights shows the q / k / LayerNorm-1 tensors AT their bf16-storage floor (5-9e-2 rel-L2): the data type's, not a kernel's.
# For scale: F2 / F4 / F10 (a matched coordinate on a kink) needed (0.155, 0.88) on every box-fed tensor and skipped box_head.* altogether.
GRAD_BANDS_TRAINED = (7.5e-2, 0.996)
MARGIN_BANDS = {
    "f2b_b16_margins": {"*": (3e-2, 0.999)},
    "f4b_l14_margins": {"*": (3.5e-2, 0.999)},
    "f2c_b16_seed_search": {"*": (3.2e-2, 0.999)},
    "f10b_b16_trained_margins": {"*": GRAD_BANDS_TRAINED},
}


@pytest.mark.parametrize("cname,tag,profile", MARGIN_FIXTURES)
def test_train_step_matches_reference_margin_fixture(golden_dir, cname, tag, profile):
    path = os.path.join(golden_dir, f"{tag}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{tag}.npz not generated")
    cfg = get_config(cname)
    g = np.load(path)
    img = synth.make_images(cfg, 1)
    labels, boxes = [g["tgt_labels"]], [g["tgt_boxes"]]
    model, crit, lg, grads, pb, ps = _step_hip(cfg, weights.make_weights(cfg, profile=profile), img, labels, boxes, g["scales"])
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    n = len(labels[0])
    same_tc = bool((crit.last["target_classes"][0].cpu() == torch.from_numpy(g["target_classes"])).all())
    same_idx = np.array_equal(crit.last["pred_idx"][0, :n].cpu().numpy(), g["pred_idx"]) and np.array_equal(crit.last["tgt_idx"][0, :n].cpu().numpy(), g["tgt_idx"])
    print(f"{tag}: max|d boxes|={eb:.3e} max|d sims|={es:.3e}; margins of the fixture: " + ", ".join(f"{k} {float(g['margin/' + k]):.3e}" for k in ("coord", "inter", "gap", "iou", "simpos"))
          + f"; assignment identical {same_idx}, labels after spreading identical {same_tc}")
    trained = profile != "init"
    assert eb < (TOL_TRAINED_BOXES if trained else TOL_BOXES) and es < (TOL_TRAINED_SIMS if trained else TOL_SIMS), (eb, es)
    assert not _near_tie(dict(pred_boxes=g["pred_boxes"], pred_idx=g["pred_idx"], tgt_idx=g["tgt_idx"]), boxes)
    seed_search = tag.startswith("f2c")
    if not seed_search:
        assert same_idx and same_tc          # a runner-up 10x the forward error away: the decisions cannot differ
    if not (same_idx and same_tc):           # (F2c only: runner-up gap below the bar -- losses like with like, gradients not comparable)
        ref_l, n_swaps, n_rows = _reference_losses_for_decisions(cfg, g, labels, boxes, crit, eb, es)
        for k in LOSS_KEYS:
            assert abs(lg[k] - ref_l[k]) <= LOSS_REL * abs(ref_l[k]), (k, lg[k], ref_l[k])
        pytest.skip(f"{tag}: {n_swaps} assignment swaps across a cost near-tie (runner-up gap {float(g['margin/gap']):.2e}): gradients not comparable")
    # losses.  Class terms: LOSS_REL, or the first-order bound for the measured max|d sims| where the -w / |sim| slope makes that larger (trained-like |sims|).
    # Box terms: the reference's value moved by the first-order effect of the matched rows' own forward deviation (like with like), LOSS_REL + 15 % of that effect.
    cbound, bfo = _class_loss_bound(cfg, g, es), _box_loss_first_order(g, boxes, pb)
    for k in LOSS_KEYS:
        ref = float(g[k])
        if k in bfo:
            pred, scale = bfo[k]
            print(f"   {k}: {lg[k]:.6f} ref {ref:.6f} rel {abs(lg[k] - ref) / abs(ref):.2e}; first-order effect of the matched rows' forward deviation {pred:+.3e} -> residual {abs(lg[k] - ref - pred) / abs(ref):.2e}")
            assert abs(lg[k] - ref - pred) <= LOSS_REL * abs(ref) + 0.15 * scale, (k, lg[k], ref, pred, scale)
        else:
            print(f"   {k}: {lg[k]:.6f} ref {ref:.6f} rel {abs(lg[k] - ref) / abs(ref):.2e} (first-order bound {cbound[k] / abs(ref):.2e})")
            assert abs(lg[k] - ref) <= max(LOSS_REL * abs(ref), cbound[k] if trained else 0.0), (k, lg[k], ref, cbound[k])
    # all 29 tensors, nothing skipped
    assert len(grads) == 29
    _check_grads_vs_fixture(g, grads, tag, MARGIN_BANDS[tag])


@pytest.mark.parametrize("cname", ["owlvit-base-patch16", "owlvit-large-patch14", "small"])
def test_loss_backward_at_the_operating_point(cname):
    """What the end-to-end fixtures cannot pin when a matched box sits on a kink of the loss: d loss / d (pred_boxes, pred_sims) of the HIP criterion
    against the oracle's autograd of the same loss evaluated AT THE HIP FORWARD'S OWN OUTPUTS (same inputs into both -> same decisions, same kinks).
    Measured: 1.2e-7 / 1.9e-6 absolute on gradients of magnitude 2.0 / 15."""
    cfg = get_config(cname)
    B = 1 if cname.startswith("owlvit") else 3
    img = synth.make_images(cfg, B)
    labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
    scales = synth.class_scales(cfg, labels)
    model = OwlViT(cfg, weights.make_weights(cfg), DEV)
    crit = PushPullLoss(cfg.n_classes, scales)
    pb, _, ps, _ = model(torch.from_numpy(img).to(DEV))
    pb.retain_grad(); ps.retain_grad()
    l = crit(ps, [torch.from_numpy(x).to(DEV) for x in labels], pb, [torch.from_numpy(x).to(DEV) for x in boxes])
    (l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
    sims = ps.detach().cpu().clone().requires_grad_(True); bx = pb.detach().cpu().clone().requires_grad_(True)
    details = []
    lo = O.push_pull_loss(sims, [torch.from_numpy(x) for x in labels], bx, [torch.from_numpy(x) for x in boxes], cfg.n_classes, torch.from_numpy(scales), details)
    (lo["loss_ce"] + lo["loss_bg"] + lo["loss_bbox"] + lo["loss_giou"]).backward()
    for b in range(B):
        n = len(labels[b])
        assert np.array_equal(crit.last["pred_idx"][b, :n].cpu().numpy(), details[b]["pred_idx"].numpy())
        assert np.array_equal(crit.last["tgt_idx"][b, :n].cpu().numpy(), details[b]["tgt_idx"].numpy())
        assert np.array_equal(crit.last["target_classes"][b].cpu().numpy(), details[b]["target_classes"].numpy())
    for k in LOSS_KEYS:
        assert float(l[k]) == pytest.approx(float(lo[k]), rel=2e-5, abs=1e-6), k
    db, ds = float((pb.grad.cpu() - bx.grad).abs().max()), float((ps.grad.cpu() - sims.grad).abs().max())
    print(f"{cname}: d_boxes max abs diff {db:.3e} (|.| max {float(bx.grad.abs().max()):.3e}), d_sims {ds:.3e} (|.| max {float(sims.grad.abs().max()):.3e})")
    assert db <= 1e-5 * max(1.0, float(bx.grad.abs().max())) and ds <= 1e-5 * max(1.0, float(sims.grad.abs().max())), (db, ds)


def test_l14_train_step_matches_reference_fixture_f4(golden_dir):
    """BASELINE configs[4] architecture (owlvit-large-patch14, 840x840) at batch 1: patch 14 (gathered: rows padded to 16 positions in the K index),
    T = 3601, and the literal `layers.11` rule on 24 layers -> backward through 12 frozen layers."""
    path = os.path.join(golden_dir, "f4_l14.npz")
    if not os.path.exists(path):
        pytest.skip("f4 fixture missing")
    cfg = get_config("owlvit-large-patch14")
    g = np.load(path)
    img = synth.make_images(cfg, 1)
    labels, boxes = synth.make_targets(cfg, 1, max_boxes=16)
    model, crit, lg, grads, pb, ps = _step_hip(cfg, weights.make_weights(cfg), img, labels, boxes, g["scales"])
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    same = float((crit.last["target_classes"][0].cpu() == torch.from_numpy(g["target_classes"])).float().mean())
    print(f"L/14 vs reference fixture: max|d boxes|={eb:.3e} max|d sims|={es:.3e}; losses", lg,
          "ref", {k: float(g[k]) for k in LOSS_KEYS}, "target agreement", same)
    assert eb < TOL_BOXES and es < TOL_SIMS, (eb, es)
    ref_l, n_swaps, n_rows = _reference_losses_for_decisions(cfg, g, labels, boxes, crit, eb, es)
    print("near-tie decisions differing from the fixture: assignment swaps", n_swaps, "label rows", n_rows,
          "-> reference losses under these decisions:", ref_l)
    for k in LOSS_KEYS:
        ref = ref_l[k] if (n_swaps or n_rows) else float(g[k])
        print(f"   {k}: {lg[k]:.6f} ref {ref:.6f} rel {abs(lg[k] - ref) / abs(ref):.2e}")
        assert abs(lg[k] - ref) <= LOSS_REL * abs(ref), (k, lg[k], ref)
    near_tie = _near_tie(g, boxes)      # (here row 610: x1 = 0.1997 vs 0.1996)
    print("near-tie between a matched prediction and its target:", near_tie)
    if not (n_swaps or n_rows):         # (other decisions: other gradients -- the losses above are compared like with like, the gradients cannot be)
        _check_grads_vs_fixture(g, grads, "L/14 F4", GRAD_BANDS_F4, skip=lambda n: near_tie and n.startswith("box_head"))


# ---------------------------------------------------------------------------------------------------
# F10: trained-like statistics end to end against the reference.  Every other fixture is taken at HF-init random weights
# (near-uniform softmax, no outlier channels, |sims| <~ 0.2); weights.make_weights(profile="trained_like") has massive residual channels, LayerNorm
# gains over two decades, attention logits of std ~ 8 with sink keys that trip the forward kernel's stale-offset verdict, and |sims| > 0.9.
# The north star's bf16 bar (outputs within 1e-2) is the assertion; measured values are printed and quoted in DESIGN.md.
# ---------------------------------------------------------------------------------------------------
TOL_TRAINED = 1e-2                                  # the north star's bar
# ... asserted at ~2x the measured deviation of the round-4 build (profiles/r04_parity_bands.md): "trained_like" B/16 boxes 4.0e-3 (rms 7.5e-4) /
# sims 1.6e-4, tiny 5.1e-4 / 5.2e-5; slow-path tiles 906 against 930 predicted from the reference's own logits
TOL_TRAINED_BOXES, TOL_TRAINED_SIMS = 8e-3, 4e-4


def _sample(gr, n=4096):
    flat = gr.reshape(-1)
    stride = max(1, flat.numel() // n)
    return flat[::stride][:n]


@pytest.mark.parametrize("cname,tag,max_boxes,profile", [("tiny", "f10_tiny_trained", 6, "trained_like"), ("owlvit-base-patch16", "f10_b16_trained", 16, "trained_like"),
                                                         ("owlvit-base-patch16", "f10_b16_trained_hard", 16, "trained_like_hard")])
def test_trained_like_train_step_matches_reference_fixture_f10(golden_dir, cname, tag, max_boxes, profile):
    from owl_vit_object_detection_amd import ops
    hard = profile.endswith("hard")
    cfg = get_config(cname)
    g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
    img = synth.make_images(cfg, 1)
    labels, boxes = synth.make_targets(cfg, 1, max_boxes=max_boxes)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.ATTN_SLOW_TILES = counter
    try:
        Wnp = weights.make_weights(cfg, profile=profile)
        model, crit, lg, grads, pb, ps = _step_hip(cfg, Wnp, img, labels, boxes, g["scales"])
        torch.cuda.synchronize()
    finally:
        ops.ATTN_SLOW_TILES = None
    slow = int(counter.item())
    eb, es = _maxerr(pb, torch.from_numpy(g["pred_boxes"])), _maxerr(ps, torch.from_numpy(g["pred_sims"]))
    rb = float((pb.cpu() - torch.from_numpy(g["pred_boxes"])).pow(2).mean().sqrt()); rs = float((ps.cpu() - torch.from_numpy(g["pred_sims"])).pow(2).mean().sqrt())
    same = float((crit.last["target_classes"][0].cpu() == torch.from_numpy(g["target_classes"])).float().mean())
    print(f"F10 {cname}: max|d boxes|={eb:.3e} (rms {rb:.2e}) max|d sims|={es:.3e} (rms {rs:.2e}); max|sims| ref {float(np.abs(g['pred_sims']).max()):.3f}; "
          f"slow-path tiles: kernel {slow}, predicted from the reference's logits {int(g['attn/slow_tiles'].sum())}; target agreement {same}")
    print("   losses", lg, "ref", {k: float(g[k]) for k in LOSS_KEYS})
    if not hard:
        assert eb < TOL_TRAINED_BOXES and es < TOL_TRAINED_SIMS, (eb, es)
    else:
        # Two decades of LayerNorm gain make the network ill-conditioned: bf16 STORAGE alone (the fp32 oracle with the HIP path's rounding points and
        # nothing else changed, tests/bf16_emulation.py) misses the 1e-2 bar on pred_boxes.  What is asserted: sims hold the bar, and the HIP path's
        # boxes are no further from the reference than that data-type floor allows (1.5x: two roundings of the same size do not coincide).
        from tests.bf16_emulation import model_forward_bf16_storage
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
        with torch.no_grad():
            emb, ems = model_forward_bf16_storage(cfg, w, torch.from_numpy(img))
        fb = float((emb - torch.from_numpy(g["pred_boxes"])).abs().max()); frb = float((emb - torch.from_numpy(g["pred_boxes"])).pow(2).mean().sqrt())
        print(f"   bf16-storage floor (emulation vs reference): boxes max {fb:.3e} rms {frb:.3e}; HIP vs emulation max {_maxerr(pb, emb):.3e}")
        assert es < TOL_TRAINED and fb > TOL_TRAINED, (es, fb)
        assert eb < 1.5 * fb and rb < 1.5 * frb, (eb, fb, rb, frb)
        # ... and ABSOLUTE bands (round 6), ~1.4x the measured 3.49e-2 / 4.75e-3: the floor-relative assertion above cannot move with a
        # regression of the emulation itself.  Why no compensated path brings this weight set under 1e-2 at < 10 % cost: profiles/r06_hard_localise.md
        # (every one of the 14 bf16 storage points alone moves the boxes 9e-4 ... 1.6e-2 -- the input cast alone 1.5e-2).
        assert eb < 5e-2 and rb < 7e-3, (eb, rb)
    # the attention forward's stale-offset verdict: exercised where the reference's own logits say it must be (and only there)
    pred = int(g["attn/slow_tiles"].sum())
    if pred == 0:
        assert slow == 0
    else:                       # (a tile whose row sum sits within bf16 noise of 2^88 may land on either side)
        assert abs(slow - pred) <= max(0.25 * pred, 8), (slow, pred)
    bound = _class_loss_bound(cfg, g, es)
    ref_l, n_swaps, n_rows = _reference_losses_for_decisions(cfg, g, labels, boxes, crit, eb, es)
    print("   decisions differing from the fixture: assignment swaps", n_swaps, "label rows", n_rows)
    for k in LOSS_KEYS:
        ref = ref_l[k] if (n_swaps or n_rows) else float(g[k])
        print(f"   {k}: {lg[k]:.6f} ref {ref:.6f} rel {abs(lg[k] - ref) / abs(ref):.2e}")
        # (hard profile: matched boxes move by up to the data-type floor above, 4e-2 -- the box losses follow: measured 2.1 % on loss_giou)
        assert abs(lg[k] - ref) <= max((6e-2 if hard else 2e-2) * abs(ref), 1e-2, bound.get(k, 0.0)), (k, lg[k], ref, bound)
    # gradients: norm ratio + cosine on a 4096-element strided sample of every tensor that carries signal.  As in F2, a matched row of the B/16 fixture
    # sits on a kink of the box loss (row 777: both x edges within 1.3e-3 of its target's, |d_box| 1.5 against 0.1-0.7 for the other twelve --
    # gpurun_out/r4_diag_box_f10.log): tensors fed by the class loss only are held tight, box-fed ones to the F2 band, box_head.* is pinned by
    # test_loss_backward_at_the_operating_point + the chain test below instead.
    if hard or n_swaps or n_rows:
        return
    near_tie = _near_tie(g, boxes)
    full = "grad/queries" in g.files
    big = max(float(np.linalg.norm(g["grad/" + n])) if full else float(g["gradnorm/" + n]) for n in grads)
    worst_norm, worst_cos, lines, bad = 0.0, 1.0, [], []
    for n, gr in grads.items():
        ref_full = torch.from_numpy(g["grad/" + n]) if full else None
        ref_norm = float(ref_full.double().norm()) if full else float(g["gradnorm/" + n])
        ref_s = _sample(ref_full) if full else torch.from_numpy(g["gradsample/" + n])
        ours_s = _sample(gr)
        ratio = float(gr.double().norm()) / max(ref_norm, 1e-30)
        cos = float((ours_s.double() * ref_s.double()).sum() / (ours_s.double().norm() * ref_s.double().norm() + 1e-30))
        lines.append(f"     {n:58s} |ref|={ref_norm:.3e} norm ratio {ratio:.4f} cos(sample) {cos:.5f}")
        if ref_norm < 1e-2 * big or (near_tie and n.startswith("box_head")):
            continue
        bn, bc = (5e-3, 0.9999) if n in _CLASS_ONLY else (GRAD_BANDS_F2_BOX_FED if near_tie else (2e-2, 0.995))
        worst_norm = max(worst_norm, abs(ratio - 1.0)); worst_cos = min(worst_cos, cos)
        if abs(ratio - 1.0) > bn or cos < bc:
            bad.append((n, ratio, cos, bn, bc))
    print("\n".join(lines))
    print(f"   F10 {cname} end-to-end gradients (near-tie on a box-loss kink: {near_tie}): worst |norm ratio - 1| = {worst_norm:.3e}, worst sample cosine = {worst_cos:.5f}")
    assert not bad, bad


@pytest.mark.parametrize("cname,B", [("tiny", 2), ("owlvit-base-patch16", 1)])
def test_backward_chain_trained_like_matches_oracle_given_same_upstream(cname, B):
    """The backward-chain test on trained-like weights: peaked softmax through the attention backward, large LayerNorm gains through ln_bwd."""
    cfg = get_config(cname)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Wnp = weights.make_weights(cfg, profile="trained_like")
    img = synth.make_images(cfg, B)
    gen = torch.Generator().manual_seed(5)
    d_boxes = torch.randn(B, cfg.patches, 4, generator=gen) * 0.1
    d_sims = torch.randn(B, cfg.patches, cfg.n_classes, generator=gen) * 0.1
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    names = O.trainable_names(w)
    ww = {n: (t.clone().requires_grad_(True) if n in names else t) for n, t in w.items()}
    taps = {}
    rb, rs = O.model_forward(cfg, ww, torch.from_numpy(img), taps)
    with torch.no_grad():
        e = torch.nn.functional.linear(taps["feats"], w["class_predictor.dense0.weight"], w["class_predictor.dense0.bias"])
        e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
        q = w["queries"] / torch.linalg.norm(w["queries"], dim=-1, keepdim=True) + 1e-6
        top2 = (e @ q.transpose(1, 2)).view(B, cfg.patches, cfg.n_classes, 3).topk(2, dim=-1).values
        d_sims = d_sims * ((top2[..., 0] - top2[..., 1]) > 0.02).float()
    model = OwlViT(cfg, Wnp, DEV)
    pb, _, ps, _ = model(torch.from_numpy(img).to(DEV))
    torch.autograd.backward([pb, ps], [d_boxes.to(DEV), d_sims.to(DEV)])
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}
    torch.autograd.backward([rb, rs], [d_boxes, d_sims])
    gref = {n: ww[n].grad for n in names}
    worst, worst_cos = _grad_report(grads, gref, f"backward-only trained-like {cname} B={B}")
    # measured (profiles/r04_parity_bands.md): tiny 3.2e-2 / 0.99956; B/16 1.0e-1 / 0.99790, worst on layer 11's layer_norm1.weight (|ref| 0.19 beside
    # v_proj.weight's 10.0) and its q / k projections (4-5e-2): P and dS in bf16 under a peaked softmax (logit std 8-10, sink keys); at HF-init weights
    # the same chain measures <= 9.5e-3 (test_backward_chain_matches_oracle_given_same_upstream)
    assert worst < 0.2 and worst_cos > 0.995, (worst, worst_cos)
    # ... and that this IS the data type: the same backward with nothing but the HIP path's bf16 STORAGE points -- forward and backward:
    # P / dS as bf16 MFMA operands recomputed from the forward's log-sum-exp, bf16 gradients of every stored activation, activation derivatives at the
    # stored pre-activations (tests/bf16_emulation.py: model_forward_bf16_storage_train) -- sits as far from the fp32 oracle, tensor by tensor.  A kernel
    # error would show as a tensor where HIP is well outside its floor.
    from tests.bf16_emulation import model_forward_bf16_storage_train
    we = {n: (t.clone().requires_grad_(True) if n in names else t) for n, t in w.items()}
    eb_, es_ = model_forward_bf16_storage_train(cfg, we, torch.from_numpy(img))
    torch.autograd.backward([eb_, es_], [d_boxes, d_sims])
    floor_scale = 1e-3 * max(float(r.float().norm()) for r in gref.values())
    lines, bad = [], []
    for n in names:
        r = gref[n].float()
        den = max(float(r.norm()), floor_scale)
        e_hip, e_emu, e_he = float((grads[n] - r).norm()) / den, float((we[n].grad - r).norm()) / den, float((grads[n] - we[n].grad).norm()) / den
        lines.append(f"  {n:58s} HIP vs oracle {e_hip:.3e}   bf16-storage floor {e_emu:.3e}   HIP vs emulation {e_he:.3e}")
        if e_hip > 2.5 * e_emu + 5e-3:          # (HIP and the emulation round at the same points but not the same values: two independent errors of one size;
                                                #  measured: B/16 every tensor <= 1.53x its floor, layer_norm1.weight 9.0e-2 vs 5.9e-2; tiny q_proj.bias 2.5e-2 vs 1.1e-2)
            bad.append((n, e_hip, e_emu))
    print("\n".join(lines))
    assert not bad, bad


def test_load_model_from_an_hf_named_state_dict():
    """VERDICT r03 missing #3: an HF `OwlViTForObjectDetection.state_dict()` goes through `weights.from_hf_state_dict` into `load_model(..., state=)` and
    gives the very outputs of the same weights under the reference's names."""
    from owl_vit_object_detection_amd.models import load_model
    from tests.test_weights import _hf_state_dict
    cfg = get_config("tiny")
    Wnp = weights.make_weights(cfg, profile="trained_like")
    sd = _hf_state_dict(cfg, Wnp)
    state = weights.from_hf_state_dict(sd, queries=torch.from_numpy(Wnp["queries"][0]))
    labelmap = {str(i): i for i in range(cfg.n_classes)}
    a = load_model(labelmap, DEV, arch="tiny", state=state).eval()
    b = OwlViT(cfg, Wnp, DEV).eval()
    img = torch.from_numpy(synth.make_images(cfg, 2)).to(DEV)
    with torch.no_grad():
        pa, _, sa, _ = a(img)
        pb, _, sb, _ = b(img)
    assert torch.equal(pa, pb) and torch.equal(sa, sb)
    assert {n for n, p in a.named_parameters() if p.requires_grad} == {n for n in weights.param_shapes(cfg) if weights.is_trainable(n)}
