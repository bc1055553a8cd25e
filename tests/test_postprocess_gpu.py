"""-m gpu: device post-process (owl_postprocess) vs the oracle restatement and the reference-generated fixture F6
(ref src/models.py:122-146; main.py:114-117).  Index/class outputs must be bit-exact, scores/boxes bit-exact too
(they are copies of inputs)."""
import os

import numpy as np
import pytest
import torch

from oracle import owl_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(P, C, seed, dup=False, cluster=False):
    rng = np.random.default_rng(seed)
    if cluster:     # boxes in a few tight clusters -> long suppression chains
        centers = rng.random((8, 2)).astype(np.float32) * 0.6 + 0.1
        c = centers[rng.integers(0, 8, P)] + rng.normal(0, 0.01, (P, 2)).astype(np.float32)
        wh = np.float32(0.2) + rng.normal(0, 0.01, (P, 2)).astype(np.float32)
    else:
        c = rng.random((P, 2)).astype(np.float32) * 0.7
        wh = rng.random((P, 2)).astype(np.float32) * 0.3 + np.float32(0.02)
    boxes = np.concatenate([c, c + wh], 1).astype(np.float32)
    sims = (rng.random((P, C)).astype(np.float32) * 2 - 1) * np.float32(0.6)
    if dup:         # exact score ties + identical boxes
        boxes[1::7] = boxes[0:-1:7][: len(boxes[1::7])]
        sims[1::7] = sims[0:-1:7][: len(sims[1::7])]
    return boxes, sims


ROUTES = ("per_class", "coordinate_offset", "torchvision_gpu", "torchvision_cpu")


def _run(boxes, sims, conf, iou, top_k=None, route="per_class"):
    from owl_vit_object_detection_amd.postprocess import PostProcess
    pp = PostProcess(conf, iou, nms_route=route)
    b = torch.from_numpy(boxes).cuda()[None]
    s = torch.from_numpy(sims).cuda()[None]
    ob, oc, os_ = pp(b, s, top_k=top_k)
    return ob[0].cpu().numpy(), oc[0].cpu().numpy(), os_[0].cpu().numpy(), pp.last_patch_idx[0].cpu().numpy()


@pytest.mark.parametrize("P,C,conf,iou,kw", [
    (36, 4, 0.0, 0.3, {}), (576, 10, 0.01, 0.6, {}), (2304, 10, 0.01, 0.6, {}), (2304, 10, 0.3, 0.45, dict(cluster=True)),
    (3600, 10, 0.01, 0.6, dict(cluster=True)), (2304, 3, -1.0, 0.3, dict(dup=True)), (5000, 7, 0.0, 0.5, {}),
    (100, 5, 0.99, 0.5, {}),     # nothing passes the threshold
])
@pytest.mark.parametrize("route", ROUTES)
def test_postprocess_vs_oracle(P, C, conf, iou, kw, route):
    """Each of torchvision's two batched_nms routes (and its own size / device dispatch between them) against the oracle's restatement of it."""
    boxes, sims = _case(P, C, seed=P + C, **kw)
    eb, ec, es, ei = O.post_process(boxes, sims, conf, iou, route=route)
    ob, oc, os_, oi = _run(boxes, sims, conf, iou, route=route)
    assert oi.shape == ei.shape, (oi.shape, ei.shape)
    assert np.array_equal(oi, ei)
    assert np.array_equal(oc, ec)
    assert np.array_equal(os_, es) and np.array_equal(ob, eb)
    assert oc.dtype == np.int64
    if len(es) > 1:
        assert np.all(np.diff(es) <= 0)            # score-descending, like batched_nms


def test_postprocess_topk_is_prefix():
    boxes, sims = _case(2304, 10, seed=5)
    _, _, _, full = _run(boxes, sims, 0.01, 0.6)
    _, _, s200, top = _run(boxes, sims, 0.01, 0.6, top_k=200)     # (per_class route on both sides)
    assert len(top) == min(200, len(full)) and np.array_equal(top, full[:200])
    # == torch.topk(scores, min(200, K)) of the un-truncated result (ref main.py:114-117)
    eb, ec, es, ei = O.post_process(boxes, sims, 0.01, 0.6, top_k=200)
    assert np.array_equal(top, ei) and np.array_equal(s200, es)


def test_postprocess_batched_matches_per_image():
    from owl_vit_object_detection_amd.postprocess import PostProcess
    cases = [_case(2304, 10, seed=s, cluster=(s % 2 == 0)) for s in range(5)]
    b = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    s = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    pp = PostProcess(0.01, 0.6, nms_route="per_class")
    ob, oc, os_ = pp(b, s, top_k=200)
    counts = pp.last_counts.cpu().numpy()
    assert ob.shape == (5, 200, 4) and oc.shape == (5, 200)
    for i, (bx, sm) in enumerate(cases):
        eb, ec, es, ei = O.post_process(bx, sm, 0.01, 0.6, top_k=200)
        k = counts[i]
        assert k == len(ei)
        assert np.array_equal(pp.last_patch_idx[i, :k].cpu().numpy(), ei)
        assert np.array_equal(oc[i, :k].cpu().numpy(), ec)
        assert np.all(oc[i, k:].cpu().numpy() == -1)


@pytest.mark.parametrize("route,tag", [("per_class", "out"), ("coordinate_offset", "off_out")])
def test_postprocess_fixture_f6(route, tag):
    """Outputs of the REFERENCE's PostProcess run in the build container (tests/golden/make_golden.py f6), once per torchvision route; the last
    case holds ~1150 same-class pairs whose IoU sits within a few ulps of the threshold -- where the two routes part (26 more boxes kept
    on shifted coordinates)."""
    z = np.load(os.path.join(GOLD, "f6_postprocess.npz"))
    n = int(z["n_cases"])
    for k in range(n):
        boxes, sims = z[f"boxes_{k}"], z[f"sims_{k}"]
        conf, iou = float(z[f"conf_{k}"]), float(z[f"iou_{k}"])
        ob, oc, os_, _ = _run(boxes, sims, conf, iou, route=route)
        assert np.array_equal(oc, z[f"{tag}_classes_{k}"][0])
        assert np.array_equal(os_, z[f"{tag}_scores_{k}"][0])
        assert np.array_equal(ob, z[f"{tag}_boxes_{k}"][0])
    # the default route is torchvision's choice for a GPU tensor: the coordinate trick at these sizes
    k = n - 1
    ob, oc, os_, _ = _run(z[f"boxes_{k}"], z[f"sims_{k}"], float(z[f"conf_{k}"]), float(z[f"iou_{k}"]), route="torchvision_gpu")
    assert np.array_equal(os_, z[f"off_out_scores_{k}"][0]) and not np.array_equal(os_.shape, z[f"out_scores_{k}"][0].shape)


def test_postprocess_on_model_outputs():
    """End to end through the reference call surface (ref main.py:110-117) on the tiny model."""
    from owl_vit_object_detection_amd import synth
    from owl_vit_object_detection_amd.models import PostProcess, load_model
    model = load_model({str(i): i for i in range(4)}, "cuda", arch="tiny").eval()
    img = torch.from_numpy(synth.make_images(model.cfg, 1, seed=3)).cuda()
    with torch.no_grad():
        pred_boxes, _, pred_sims, _ = model(img)
    pp = PostProcess(confidence_threshold=0.01, iou_threshold=0.6)           # default route = torchvision's on a GPU
    b, c, s = pp(pred_boxes, pred_sims)
    eb, ec, es, ei = O.post_process(pred_boxes[0].cpu().numpy(), pred_sims[0].cpu().numpy(), 0.01, 0.6, route="torchvision_gpu")
    assert b.shape[0] == 1 and b.shape[2] == 4 and c.shape == s.shape == b.shape[:2]
    assert np.array_equal(c[0].cpu().numpy(), ec) and np.array_equal(s[0].cpu().numpy(), es)
