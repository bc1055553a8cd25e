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
    assert eb < 1e-2 and es < 1e-2


def test_forward_matches_reference_fixture_f1(golden_dir):
    cfg = get_config("tiny")
    g = np.load(os.path.join(golden_dir, "f1_tiny.npz"))
    model = OwlViT(cfg, weights.make_weights(cfg), DEV)
    img = torch.from_numpy(synth.make_images(cfg, 1)).to(DEV)
    with torch.no_grad():
        pb, _, ps, _ = model(img)
    assert _maxerr(pb, torch.from_numpy(g["pred_boxes"])) < 1e-2
    assert _maxerr(ps, torch.from_numpy(g["pred_sims"])) < 1e-2


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
    assert eb < 1e-2 and es < 1e-2
    # batch invariance: image 0 inside a batch of 8 gives the same outputs
    imgs = torch.from_numpy(synth.make_images(cfg, 8)).to(DEV)
    with torch.no_grad():
        pb8, _, ps8, _ = model(imgs)
    assert _maxerr(pb8[0], pb[0]) < 1e-6 and _maxerr(ps8[0], ps[0]) < 1e-6
