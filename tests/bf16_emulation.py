"""Test infrastructure (checker side, CPU): the oracle's forward with every activation the HIP path STORES in bf16 rounded to bf16 at the same point --
and nothing else changed (f32 accumulation, f32 residual stream, f32 LayerNorm statistics, f32 softmax sums).  It answers one question: how far from the
fp32 reference does bf16 STORAGE alone put the outputs on a given weight set?  If the HIP path sits at that distance from the reference and much closer to
this emulation, its deviation is the data type's, not a kernel's.  Rounding points: bf16 weights; h = LN(x); q, k, v; the softmax
scale folded into Q and re-rounded (attention_fwd.hip); P (unnormalised, against the row maximum) as the PV operand, row sums from the f32 P; the attention
output; both branch outputs ("deltas"); the MLP hidden g; feats; the box head's two hidden layers.  Cites: ref src/models.py:98-119, HF5:377-402,463-509."""
import torch
import torch.nn.functional as F

from oracle import owl_oracle as O


SKIP = set()       # rounding points switched OFF (study only: which storage matters) -- names: w h q k qs v p o d1 g d2 feats box


def bf(x, name=None):
    if name is not None and name in SKIP:
        return x
    return x.to(torch.bfloat16).to(torch.float32)


def _layer(x, w, pre, heads, eps):
    B, T, D = x.shape
    dh = D // heads
    lin = lambda t, n: F.linear(t, bf(w[pre + n + ".weight"], "w"), w[pre + n + ".bias"])
    h = bf(F.layer_norm(x, (D,), w[pre + "layer_norm1.weight"], w[pre + "layer_norm1.bias"], eps), "h")
    q = bf(lin(h, "self_attn.q_proj"), "q").view(B, T, heads, dh).transpose(1, 2)
    k = bf(lin(h, "self_attn.k_proj"), "k").view(B, T, heads, dh).transpose(1, 2)
    v = bf(lin(h, "self_attn.v_proj"), "v").view(B, T, heads, dh).transpose(1, 2)
    c = (dh ** -0.5) * 1.4426950408889634
    s = torch.matmul(bf(q * c, "qs"), k.transpose(2, 3))                 # log2 domain
    p = torch.exp2(s - s.max(-1, keepdim=True).values)
    o = torch.matmul(bf(p, "p"), v) / p.sum(-1, keepdim=True)
    o = bf(o.transpose(1, 2).reshape(B, T, D), "o")
    x = x + bf(lin(o, "self_attn.out_proj"), "d1")
    h2 = bf(F.layer_norm(x, (D,), w[pre + "layer_norm2.weight"], w[pre + "layer_norm2.bias"], eps), "h")
    g = bf(O.quick_gelu(lin(h2, "mlp.fc1")), "g")
    return x + bf(lin(g, "mlp.fc2"), "d2")


def model_forward_bf16_storage(cfg, w, image, taps=None):
    D, eps, g = cfg.hidden, cfg.ln_eps, cfg.grid
    B = image.shape[0]
    pe = F.conv2d(bf(image), bf(w["backbone.embeddings.patch_embedding.weight"], "w"), stride=cfg.patch_size).flatten(2).transpose(1, 2)
    cls = w["backbone.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + w["backbone.embeddings.position_embedding.weight"].unsqueeze(0)
    x = F.layer_norm(x, (D,), w["backbone.pre_layernorm.weight"], w["backbone.pre_layernorm.bias"], eps)
    for i in range(cfg.layers):
        x = _layer(x, w, f"backbone.encoder.layers.{i}.", cfg.heads, eps)
        if taps is not None:
            taps[f"backbone.encoder.layers.{i}.out"] = x
    x = F.layer_norm(x, (D,), w["backbone.post_layernorm.weight"], w["backbone.post_layernorm.bias"], eps)
    x = x[:, 1:, :] * x[:, :1, :]
    feats = bf(F.layer_norm(x, (D,), w["post_post_layernorm.weight"], w["post_post_layernorm.bias"], eps), "feats")
    if taps is not None:
        taps["feats"] = feats
    b = bf(F.gelu(F.linear(feats, bf(w["box_head.dense0.weight"], "w"), w["box_head.dense0.bias"])), "box")
    b = bf(F.gelu(F.linear(b, bf(w["box_head.dense1.weight"], "w"), w["box_head.dense1.bias"])), "box")
    b = F.linear(b, w["box_head.dense2.weight"], w["box_head.dense2.bias"])
    b = torch.sigmoid(b + O.box_bias(g).to(b.dtype))
    cx, cy, bw_, bh = b.unbind(-1)
    boxes = torch.stack([cx - 0.5 * bw_, cy - 0.5 * bh, cx + 0.5 * bw_, cy + 0.5 * bh], dim=-1)
    e = F.linear(feats, bf(w["class_predictor.dense0.weight"], "w"), w["class_predictor.dense0.bias"])
    e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
    q = w["queries"] / torch.linalg.norm(w["queries"], dim=-1, keepdim=True) + 1e-6
    sims = F.max_pool1d(e @ q.transpose(1, 2), kernel_size=3, stride=3)
    return boxes, sims
