"""Query-bank initialisation (ref src/models.py:155-169).  CPU: oracle.text_forward vs HF outputs
(fixture F8, produced by HF OwlViTForObjectDetection(...).text_embeds in the build container).  GPU: the device
TextTower vs the same fixture (bf16 MFMA linears: tolerance 1e-2 on unit-norm rows, stated below)."""
import os

import numpy as np
import pytest
import torch

from oracle import owl_oracle as O
from owl_vit_object_detection_amd import weights
from owl_vit_object_detection_amd.config import get_text_config

CASES = ["tiny", "owlvit-base-patch16"]


@pytest.mark.parametrize("cname", CASES)
def test_text_oracle_vs_hf_fixture(golden_dir, cname):
    z = np.load(os.path.join(golden_dir, "f8_text_embeds.npz"))
    tc = get_text_config(cname)
    got = O.text_forward(tc, weights.make_text_weights(tc), z[cname + "/input_ids"]).numpy()
    ref = z[cname + "/text_embeds"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5


def test_padding_after_eos_is_irrelevant():
    """The device tower applies the causal mask only: ids after the EOS token must not change the pooled row."""
    tc = get_text_config("tiny")
    w = weights.make_text_weights(tc)
    ids = np.array([[95, 4, 9, 96, 0, 0, 0, 0], [95, 4, 9, 96, 7, 7, 3, 1]], np.int64)
    e = O.text_forward(tc, w, ids).numpy()
    assert np.abs(e[0] - e[1]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cname", CASES)
def test_device_text_tower_vs_hf_fixture(golden_dir, cname):
    from owl_vit_object_detection_amd.text import TextTower
    z = np.load(os.path.join(golden_dir, "f8_text_embeds.npz"))
    tc = get_text_config(cname)
    tower = TextTower(tc)
    ids = z[cname + "/input_ids"]
    got = tower.encode(ids).cpu().numpy()
    ref = z[cname + "/text_embeds"]
    assert got.shape == ref.shape
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    assert np.abs(got - ref).max() < 1e-2                       # north_star tolerance for bf16 compute
    cos = (got * ref).sum(1)
    assert cos.min() > 0.9995, cos.min()
    # the prompt-specific part (rows minus their mean) must agree too, not only the shared component
    gc, rc = got - got.mean(0), ref - ref.mean(0)
    ccos = (gc * rc).sum(1) / (np.linalg.norm(gc, axis=1) * np.linalg.norm(rc, axis=1))
    assert ccos.min() > 0.99, ccos.min()
    qb = tower.query_bank(ids)
    assert qb.shape == (1, ids.shape[0], tc.proj_dim)


@pytest.mark.gpu
def test_load_model_with_text_queries():
    """ref src/models.py:155-169: load_model initialises `queries` from the text tower when prompt ids are given."""
    from owl_vit_object_detection_amd.models import load_model
    from owl_vit_object_detection_amd.text import TextTower
    tc = get_text_config("tiny")
    ids = np.zeros((12, 16), np.int64)
    rng = np.random.default_rng(0)
    for n in range(12):
        L = 1 + n % 6
        ids[n, 0] = 95; ids[n, 1:1 + L] = rng.integers(1, 95, L); ids[n, 1 + L] = 96
    model = load_model({str(i): i for i in range(4)}, "cuda", arch="tiny", prompt_ids=ids)
    exp = TextTower(tc).query_bank(ids)
    assert model.queries.shape == (1, 12, 64) and model.queries.requires_grad
    assert torch.equal(model.queries.detach(), exp)
