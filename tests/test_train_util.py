"""Input contract: xywh absolute pixels -> xyxy normalised, as reference src/train_util.py:4-13."""
import numpy as np
import torch

from owl_vit_object_detection_amd.train_util import coco_to_model_input, model_output_to_image


def test_coco_to_model_input_matches_hand_restatement():
    boxes = torch.tensor([[[10.0, 20.0, 30.0, 40.0], [0.0, 0.0, 640.0, 480.0], [5.5, 6.5, 1.0, 2.0]]])
    meta = {"width": torch.tensor([640.0]), "height": torch.tensor([480.0])}
    out = coco_to_model_input(boxes, meta)
    ref = np.array([[[10 / 640, 20 / 480, 40 / 640, 60 / 480], [0, 0, 1, 1], [5.5 / 640, 6.5 / 480, 6.5 / 640, 8.5 / 480]]], dtype=np.float32)
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-6)
    assert torch.equal(boxes, torch.tensor([[[10.0, 20.0, 30.0, 40.0], [0.0, 0.0, 640.0, 480.0], [5.5, 6.5, 1.0, 2.0]]]))  # no in-place
    back = model_output_to_image(out, meta)
    np.testing.assert_allclose(back.numpy()[0, 0], [10, 20, 40, 60], rtol=1e-5)


def test_batched_metadata():
    boxes = torch.rand(3, 5, 4) * 100
    meta = {"width": torch.tensor([100.0, 200.0, 400.0]), "height": torch.tensor([50.0, 100.0, 200.0])}
    out = coco_to_model_input(boxes, meta)
    for b in range(3):
        w, h = float(meta["width"][b]), float(meta["height"][b])
        x, y, bw, bh = boxes[b].unbind(-1)
        ref = torch.stack([x / w, y / h, (x + bw) / w, (y + bh) / h], -1)
        torch.testing.assert_close(out[b], ref)


def test_input_contract_matches_reference_fixture_f9(golden_dir):
    """F9 = the reference's own `coco_to_model_input` / `model_output_to_image` (ref src/train_util.py:4-24, src/util.py:83-93,
    123-129) run on DataLoader-shaped inputs in the build container (tests/golden/make_golden.py f9): bit-exact f32."""
    import os
    g = np.load(os.path.join(golden_dir, "f9_input_contract.npz"))
    cases = sorted({k.split("/")[0] for k in g.files})
    assert len(cases) >= 6
    for c in cases:
        boxes = torch.from_numpy(g[c + "/xywh"])
        w, h = (int(v) for v in g[c + "/wh"])
        meta = {"width": torch.tensor([w]), "height": torch.tensor([h])}          # what default_collate makes of the dataset's ints
        keep = boxes.clone()
        out = coco_to_model_input(boxes, meta)
        assert out.dtype == torch.float32 and out.shape == boxes.shape and torch.equal(boxes, keep)
        assert torch.equal(out, torch.from_numpy(g[c + "/xyxy_norm"])), c
        assert torch.equal(model_output_to_image(out, meta), torch.from_numpy(g[c + "/back"])), c
