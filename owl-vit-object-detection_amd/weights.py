"""Deterministic parameter set for the OWL-ViT vision path.

Parameter *names* are the ones ``named_parameters()`` yields on the reference wrapper
(reference src/models.py:41-62: ``backbone`` = HF ``vision_model``, ``post_post_layernorm`` = HF
``layer_norm``, ``class_predictor.dense0``, ``box_head.dense{0,1,2}``, ``queries``), because the
freeze rule is a substring test on those names (reference src/models.py:173-184).

There is no network on either box, so weights are random: matrices use the HF init scales
(HF5:529-553); biases / LayerNorm affine are drawn non-trivially (not 0 / 1) so that a kernel
which drops a bias or a gamma cannot pass the parity tests.
"""
from collections import OrderedDict

import numpy as np

from . import rng
from .config import OwlConfig

FREEZE_KEEP = ("layers.11", "box", "post_layernorm", "class_predictor", "queries")


def is_trainable(name: str) -> bool:
    """Literal restatement of the reference freeze rule (src/models.py:173-184)."""
    return any(s in name for s in FREEZE_KEEP)


def param_shapes(cfg: OwlConfig) -> "OrderedDict[str, tuple]":
    D, I, Dt, p = cfg.hidden, cfg.mlp, cfg.text_dim, cfg.patch_size
    s = OrderedDict()
    s["queries"] = (1, cfg.queries, Dt)
    s["backbone.embeddings.class_embedding"] = (D,)
    s["backbone.embeddings.patch_embedding.weight"] = (D, 3, p, p)
    s["backbone.embeddings.position_embedding.weight"] = (cfg.tokens, D)
    s["backbone.pre_layernorm.weight"] = (D,)
    s["backbone.pre_layernorm.bias"] = (D,)
    for i in range(cfg.layers):
        pre = f"backbone.encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[pre + f"self_attn.{nm}.weight"] = (D, D)
            s[pre + f"self_attn.{nm}.bias"] = (D,)
        s[pre + "layer_norm1.weight"] = (D,)
        s[pre + "layer_norm1.bias"] = (D,)
        s[pre + "mlp.fc1.weight"] = (I, D)
        s[pre + "mlp.fc1.bias"] = (I,)
        s[pre + "mlp.fc2.weight"] = (D, I)
        s[pre + "mlp.fc2.bias"] = (D,)
        s[pre + "layer_norm2.weight"] = (D,)
        s[pre + "layer_norm2.bias"] = (D,)
    s["backbone.post_layernorm.weight"] = (D,)
    s["backbone.post_layernorm.bias"] = (D,)
    s["post_post_layernorm.weight"] = (D,)
    s["post_post_layernorm.bias"] = (D,)
    s["class_predictor.dense0.weight"] = (Dt, D)
    s["class_predictor.dense0.bias"] = (Dt,)
    s["box_head.dense0.weight"] = (D, D)
    s["box_head.dense0.bias"] = (D,)
    s["box_head.dense1.weight"] = (D, D)
    s["box_head.dense1.bias"] = (D,)
    s["box_head.dense2.weight"] = (4, D)
    s["box_head.dense2.bias"] = (4,)
    return s


def _std(name: str, cfg: OwlConfig) -> float:
    D, L = cfg.hidden, cfg.layers
    if name.endswith("class_embedding"):
        return D ** -0.5
    if "patch_embedding" in name or "position_embedding" in name:
        return 0.02
    if any(k in name for k in ("q_proj.weight", "k_proj.weight", "v_proj.weight")):
        return D ** -0.5 * (2 * L) ** -0.5
    if "out_proj.weight" in name:
        return D ** -0.5
    if "fc1.weight" in name:
        return (2 * D) ** -0.5
    if "fc2.weight" in name:
        return D ** -0.5 * (2 * L) ** -0.5
    if "class_predictor.dense0.weight" in name or "box_head" in name and name.endswith("weight"):
        return D ** -0.5
    return 0.02


def make_weights(cfg: OwlConfig, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """name -> float32 ndarray; identical on every box for a given (cfg, seed)."""
    out = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        n = int(np.prod(shape))
        z = rng.normal(seed, cfg.name + "/" + name, n)
        if name == "queries":
            # text_embeds are L2-normalised rows (HF5:958,970)
            q = z.reshape(shape)
            q = q / np.linalg.norm(q, axis=-1, keepdims=True)
            out[name] = q.astype(np.float32)
        elif "layernorm" in name or "layer_norm" in name:
            if name.endswith("weight"):
                out[name] = (1.0 + 0.05 * z).reshape(shape).astype(np.float32)
            else:
                out[name] = (0.02 * z).reshape(shape).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = (0.02 * z).reshape(shape).astype(np.float32)
        else:
            out[name] = (_std(name, cfg) * z).reshape(shape).astype(np.float32)
    return out


def count_trainable(cfg: OwlConfig) -> int:
    return sum(int(np.prod(s)) for n, s in param_shapes(cfg).items() if is_trainable(n))


# ---- text tower (query-bank initialisation, ref src/models.py:155-169) ------------------------------------------------
def text_param_shapes(tc) -> "OrderedDict[str, tuple]":
    """HF names below ``owlvit.`` (``text_model.*``, ``text_projection.weight``)."""
    W, I = tc.width, tc.mlp
    s = OrderedDict()
    s["text_model.embeddings.token_embedding.weight"] = (tc.vocab, W)
    s["text_model.embeddings.position_embedding.weight"] = (tc.max_pos, W)
    for i in range(tc.layers):
        pre = f"text_model.encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[pre + f"self_attn.{nm}.weight"] = (W, W)
            s[pre + f"self_attn.{nm}.bias"] = (W,)
        s[pre + "layer_norm1.weight"] = (W,)
        s[pre + "layer_norm1.bias"] = (W,)
        s[pre + "mlp.fc1.weight"] = (I, W)
        s[pre + "mlp.fc1.bias"] = (I,)
        s[pre + "mlp.fc2.weight"] = (W, I)
        s[pre + "mlp.fc2.bias"] = (W,)
        s[pre + "layer_norm2.weight"] = (W,)
        s[pre + "layer_norm2.bias"] = (W,)
    s["text_model.final_layer_norm.weight"] = (W,)
    s["text_model.final_layer_norm.bias"] = (W,)
    s["text_projection.weight"] = (tc.proj_dim, W)
    return s


def make_text_weights(tc, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """Deterministic random text tower (no checkpoints on either box); scales follow the HF init (HF5:529-565)."""
    out = OrderedDict()
    W, L = tc.width, tc.layers
    for name, shape in text_param_shapes(tc).items():
        z = rng.normal(seed, tc.name + "/" + name, int(np.prod(shape)))
        if "layer_norm" in name:
            v = (1.0 + 0.05 * z) if name.endswith("weight") else 0.02 * z
        elif name.endswith("bias"):
            v = 0.02 * z
        elif "token_embedding" in name:
            v = 0.02 * z
        elif "position_embedding" in name:
            v = 0.01 * z
        elif any(k in name for k in ("q_proj", "k_proj", "v_proj")):
            v = (W ** -0.5) * z      # larger than the HF init so that attention is not uniform (a mask bug must show)
        elif "out_proj" in name or "fc2" in name:
            v = (W ** -0.5) * (2 * L) ** -0.5 * z
        elif "fc1" in name:
            v = (2 * W) ** -0.5 * z
        else:
            v = (W ** -0.5) * z
        out[name] = v.reshape(shape).astype(np.float32)
    return out
