"""Host logic of weights.py (CPU): the checkpoint-name adapter and the trained-like profile."""
import numpy as np
import pytest
import torch

from owl_vit_object_detection_amd import weights
from owl_vit_object_detection_amd.config import get_config


def _hf_state_dict(cfg, W):
    """The key set of HF `OwlViTForObjectDetection.state_dict()` (vision tower + heads + text-side / buffer keys the adapter must drop),
    spelled out here so the test needs neither `transformers` nor the reference: prefix table of ref src/models.py:41-62."""
    sd = {}
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
        sd[key] = torch.from_numpy(arr)
    sd["owlvit.vision_model.embeddings.position_ids"] = torch.arange(cfg.tokens)[None]
    sd["owlvit.text_model.embeddings.token_embedding.weight"] = torch.zeros(7, 3)
    sd["owlvit.logit_scale"] = torch.tensor(2.6)
    sd["owlvit.visual_projection.weight"] = torch.zeros(4, 4)
    sd["class_head.logit_shift.weight"] = torch.zeros(1, cfg.hidden)
    sd["class_head.logit_scale.bias"] = torch.zeros(1)
    return sd


@pytest.mark.parametrize("cname", ["tiny", "owlvit-base-patch32"])
def test_from_hf_state_dict_yields_the_reference_parameter_names(cname):
    cfg = get_config(cname)
    W = weights.make_weights(cfg)
    out = weights.from_hf_state_dict(_hf_state_dict(cfg, W), queries=W["queries"][0])
    assert set(out) == set(weights.param_shapes(cfg))
    for n, shape in weights.param_shapes(cfg).items():
        assert out[n].shape == tuple(shape) and out[n].dtype == np.float32, n
        np.testing.assert_array_equal(out[n], W[n])
    # without the query bank the adapter leaves it to the caller (load_model(prompt_ids=...) or an explicit bank)
    assert "queries" not in weights.from_hf_state_dict(_hf_state_dict(cfg, W))


def test_from_hf_state_dict_against_a_real_hf_module():
    transformers = pytest.importorskip("transformers")
    cfg = get_config("tiny")
    hf_cfg = transformers.OwlViTConfig(
        vision_config=dict(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                           image_size=cfg.image_size, patch_size=cfg.patch_size),
        text_config=dict(hidden_size=cfg.text_dim, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1, vocab_size=64,
                         max_position_embeddings=16),
        projection_dim=cfg.text_dim)
    hf = transformers.OwlViTForObjectDetection(hf_cfg)
    out = weights.from_hf_state_dict(hf.state_dict(), queries=np.zeros((cfg.queries, cfg.text_dim), np.float32))
    shapes = weights.param_shapes(cfg)
    assert set(out) == set(shapes)
    for n, shape in shapes.items():
        assert out[n].shape == tuple(shape), n


def test_trained_like_profile_is_deterministic_and_has_the_advertised_structure():
    cfg = get_config("small")
    a = weights.make_weights(cfg, profile="trained_like")
    b = weights.make_weights(cfg, profile="trained_like")
    base = weights.make_weights(cfg)
    assert list(a) == list(base)
    for n in a:
        np.testing.assert_array_equal(a[n], b[n])
        assert a[n].shape == base[n].shape and a[n].dtype == np.float32 and np.isfinite(a[n]).all(), n
    t = weights.TRAINED_LIKE
    ch = [int(f * cfg.hidden) for f in t["massive_channels"]]
    np.testing.assert_array_equal(a["backbone.pre_layernorm.bias"][ch], np.asarray(t["massive_bias"], np.float32))
    g = np.abs(a["backbone.encoder.layers.5.layer_norm1.weight"])
    lo, hi = t["ln_gain_log_range"]
    assert g.min() >= lo * 0.999 and g.max() <= hi * 1.001 and g.max() / g.min() > 0.3 * hi / lo
    np.testing.assert_allclose(a["backbone.encoder.layers.2.self_attn.q_proj.weight"], base["backbone.encoder.layers.2.self_attn.q_proj.weight"] * t["qk_gain"], rtol=1e-6)
    hard = weights.make_weights(cfg, profile="trained_like_hard")
    gh = np.abs(hard["backbone.encoder.layers.5.layer_norm1.weight"])
    assert gh.max() / gh.min() > 30 and gh.max() <= 10.001
    with pytest.raises(ValueError):
        weights.make_weights(cfg, profile="nope")


@pytest.mark.parametrize("ps", [14, 16, 24, 9, 32])
def test_patch_weight_gather_layout_reproduces_the_convolution(ps):
    """weights.patch_weight_gather_layout + the loader's pixel rule (csrc/gemm_pp2.hip stage_A / csrc/gemm.hip im2row_kernel: position `pos` of a padded patch row
    holds pixel min(8 * (pos // 8), ps - 8) + pos % 8) is the patch convolution (HF5:282-288) for any patch size: every pixel's weight counted exactly once."""
    from owl_vit_object_detection_amd.weights import patch_weight_gather_layout
    rs = np.random.default_rng(ps)
    D, G = 5, 3
    S = G * ps
    w = rs.standard_normal((D, 3, ps, ps))
    img = rs.standard_normal((3, S, S))
    wk = patch_weight_gather_layout(w, ps)
    psp = 8
    while psp < ps:
        psp *= 2
    assert wk.shape == (D, (3 * ps * psp + 63) // 64 * 64) and (psp != ps or np.array_equal(wk, w.reshape(D, -1)))
    for py in range(G):
        for px in range(G):
            a = np.zeros(wk.shape[1])
            for r in range(3 * ps):
                c, ky = divmod(r, ps)
                for pos in range(psp):
                    kx = min(8 * (pos // 8), ps - 8) + pos % 8
                    a[r * psp + pos] = img[c, py * ps + ky, px * ps + kx]
            ref = (w * img[:, py * ps:(py + 1) * ps, px * ps:(px + 1) * ps][None]).sum((1, 2, 3))
            np.testing.assert_allclose(wk @ a, ref, rtol=1e-12, atol=1e-12)
