"""Error behaviour at the drop-in boundary (Python exceptions / asserts of the reference surface,
integer status + message underneath): bad inputs fail loudly, nothing falls back silently."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owl_vit_object_detection_amd import _lib, ops  # noqa: E402
from owl_vit_object_detection_amd.losses import PushPullLoss  # noqa: E402
from owl_vit_object_detection_amd.matcher import HungarianMatcher, PackedTargets  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(P=64, C=4, n=(3, 1)):
    g = torch.Generator().manual_seed(5)
    B = len(n)
    sims = (torch.rand(B, P, C, generator=g) * 1.2 - 0.6).to(DEV)
    xy = torch.rand(B, P, 2, generator=g) * 0.6
    wh = 0.05 + torch.rand(B, P, 2, generator=g) * 0.3
    boxes = torch.cat([xy, xy + wh], -1).to(DEV)
    labels = [torch.randint(0, C, (k,), generator=g).to(DEV) for k in n]
    tb = []
    for k in n:
        t_xy = torch.rand(k, 2, generator=g) * 0.6
        tb.append(torch.cat([t_xy, t_xy + 0.05 + torch.rand(k, 2, generator=g) * 0.3], -1).to(DEV))
    return sims, boxes, labels, tb


def test_image_without_targets_is_rejected():
    """The reference drops images without boxes in its dataset (src/dataset.py:33); with none, loss_ce would be mean(empty)."""
    sims, boxes, labels, tb = _case()
    labels[1] = labels[1][:0]; tb[1] = tb[1][:0]
    with pytest.raises(ValueError, match="at least one target"):
        PushPullLoss(4, None)(sims, labels, boxes, tb)
    with pytest.raises(ValueError, match="at least one target"):
        PackedTargets(labels, tb, DEV)


def test_label_box_count_mismatch_is_rejected():
    sims, boxes, labels, tb = _case()
    with pytest.raises(ValueError, match="length mismatch"):
        PackedTargets(labels, [tb[0][:2], tb[1]], DEV)


def test_ragged_target_counts_round_trip():
    """1, 7 and 40 targets in one batch: padded slots never leak into the assignment."""
    sims, boxes, labels, tb = _case(P=128, C=4, n=(1, 7, 40))
    crit = PushPullLoss(4, None)
    out = crit(sims, labels, boxes, tb)
    assert all(np.isfinite(float(v)) for v in out.values())
    for b, k in enumerate((1, 7, 40)):
        pi = crit.last["pred_idx"][b, :k].cpu().numpy()
        assert len(set(pi.tolist())) == k and (np.diff(pi) > 0).all()        # k distinct predictions, ascending (scipy order)
        assert sorted(crit.last["tgt_idx"][b, :k].cpu().tolist()) == list(range(k))


def test_more_targets_than_predictions_is_rejected():
    sims, boxes, labels, tb = _case(P=8, C=4, n=(12,))
    with pytest.raises((_lib.OwlLibError, ValueError)):
        m = HungarianMatcher(4)
        m({"pred_logits": sims, "pred_boxes": boxes}, [{"labels": labels[0], "boxes": tb[0]}])
        torch.cuda.synchronize()


def test_wrong_dtype_and_layout_are_rejected_before_the_call():
    a = torch.randn(256, 128, device=DEV)                      # f32 where bf16 is required
    w = torch.randn(256, 128, device=DEV).bfloat16()
    out = torch.zeros(256, 256, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        ops.gemm(ops.EPI_BIAS_BF16, a, w, out)
    with pytest.raises(ValueError, match="contiguous"):
        ops.gemm(ops.EPI_BIAS_BF16, a.bfloat16().t(), w, out)
    with pytest.raises(ValueError, match="device tensor"):
        ops.gemm(ops.EPI_BIAS_BF16, a.bfloat16().cpu(), w, out)
    with pytest.raises(_lib.OwlLibError, match="multiple of 64"):
        ops.gemm(ops.EPI_BIAS_BF16, a.bfloat16()[:, :96].contiguous(), w[:, :96].contiguous(), out)
