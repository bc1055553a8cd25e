"""Test infrastructure (checker side, CPU): the oracle's forward with every activation the HIP path STORES in bf16 rounded to bf16 at the same point --
and nothing else changed (f32 accumulation, f32 residual stream, f32 LayerNorm statistics, f32 softmax sums).  It answers one question: how far from the
fp32 reference does bf16 STORAGE alone put the outputs on a given weight set?  If the HIP path sits at that distance from the reference and much closer to
this emulation, its deviation is the data type's, not a kernel's.  Rounding points: bf16 weights; h = LN(x); q, k, v; the softmax
scale folded into Q and re-rounded (attention_fwd.hip); P (unnormalised, against the row maximum) as the PV operand, row sums from the f32 P; the attention
output; both branch outputs ("deltas"); the MLP hidden g; feats; the box head's two hidden layers.  Cites: ref src/models.py:98-119, HF5:377-402,463-509."""
import torch
import torch.nn.functional as F

from oracle import owl_oracle as O


ROUND_LAYERS = None   # study only: encoder layers whose rounding points are ON (None = all)
SKIP = set()       # rounding points switched OFF (study only: which storage matters) -- names: w h q k qs v p o d1 g d2 feats box img


def bf(x, name=None):
    if name is not None and name in SKIP:
        return x
    return x.to(torch.bfloat16).to(torch.float32)


def _layer(x, w, pre, heads, eps, index=None):
    B, T, D = x.shape
    dh = D // heads
    if ROUND_LAYERS is not None and index is not None and index not in ROUND_LAYERS:
        bf = lambda t, name=None: t                    # this layer computes in f32 throughout (study)
    else:
        bf = globals()["bf"]
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
    pe = F.conv2d(bf(image, "img"), bf(w["backbone.embeddings.patch_embedding.weight"], "w"), stride=cfg.patch_size).flatten(2).transpose(1, 2)
    cls = w["backbone.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + w["backbone.embeddings.position_embedding.weight"].unsqueeze(0)
    x = F.layer_norm(x, (D,), w["backbone.pre_layernorm.weight"], w["backbone.pre_layernorm.bias"], eps)
    for i in range(cfg.layers):
        x = _layer(x, w, f"backbone.encoder.layers.{i}.", cfg.heads, eps, index=i)
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


# ---------------------------------------------------------------------------------------------------
# The same question for the BACKWARD: how far from the fp32 oracle's gradients does bf16 STORAGE alone put the gradients of the
# trainable tensors?  `model_forward_bf16_storage_train` is the forward above as a differentiable graph whose backward rounds where the HIP backward
# stores bf16 (owl-vit-object-detection_amd/autograd.py: backward_impl, csrc/attention_bwd.hip, csrc/gemm_common.h epilogues 8 / 9) -- and nowhere else:
#   * every stored activation's GRADIENT is stored in bf16 too (dxb / dxb2 = d(branch outputs), datt = d(attention output), dqkv, dh = d(LN outputs), du)
#     -> `rb`: round forward and backward;   feats: bf16 forward, f32 gradient (EPI_F32 / EPI_ACC_F32) -> `rf`;   e: f32 forward, bf16 gradient (de) -> `rg`;
#   * weights: bf16 compute copy, f32 gradient (split-K slabs) -> `rf` (straight-through);
#   * d(pre-activation) = bf16(acc * act'): ONE rounding after the product; erf-GELU (box head): the derivative taken at the STORED (bf16) pre-activation;
#     quick-GELU (round 6): the STORED derivative bf16(act'(f32 pre-activation)), saved by the forward epilogue;
#   * attention backward: P and dS recomputed per kernel from the forward's log-sum-exp -- dK / dV kernel: S = Q . bf16(c K)^T, dQ kernel: S = bf16(c Q) . K^T --,
#     P and dS = P (dP - D) rounded to bf16 as MFMA operands, D = rowsum(dO . O) on the stored (bf16) O, dK / dQ scaled in f32.
# A HIP gradient that sits as far from the oracle as this emulation does (and closer to the emulation) carries the data type's error, not a kernel's.
# ---------------------------------------------------------------------------------------------------
class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(torch.float32) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(torch.float32) if ctx.bwd else g), None, None


def rb(x): return _Round.apply(x, True, True)
def rf(x): return _Round.apply(x, True, False)
def rg(x): return _Round.apply(x, False, True)
def _r(x): return x.to(torch.bfloat16).to(torch.float32)


class _ActStore(torch.autograd.Function):
    """y = act(u).  erf-GELU (box head): backward d(u) = bf16(dy * act'(bf16(u))) -- gemm_common.h EPI_DGELU, aux = the bf16 pre-activation.
    quick-GELU (encoder MLP, round 6): the forward epilogue saves bf16(act'(u)) taken at the f32 pre-activation and the backward epilogue is one multiply:
    d(u) = bf16(dy * bf16(act'(u))) -- EPI_QGELU's aux / EPI_DQGELU."""
    @staticmethod
    def forward(ctx, u, kind):
        ctx.kind = kind
        if kind == "quick":
            s = torch.sigmoid(1.702 * u)
            ctx.save_for_backward(_r(s * (1.0 + 1.702 * u * (1.0 - s))))
        else:
            ctx.save_for_backward(_r(u))
        return O.quick_gelu(u) if kind == "quick" else F.gelu(u)

    @staticmethod
    def backward(ctx, dy):
        (sb,) = ctx.saved_tensors
        if ctx.kind == "quick":
            d = sb
        else:
            d = 0.5 * (1.0 + torch.erf(sb * 0.7071067811865476)) + sb * torch.exp(-0.5 * sb * sb) * 0.3989422804014327
        return _r(dy * d), None


class _AttnStore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        c = scale * 1.4426950408889634
        s = torch.matmul(_r(q * c), k.transpose(2, 3))
        m = s.max(-1, keepdim=True).values
        p = torch.exp2(s - m)
        l = p.sum(-1, keepdim=True)
        o = torch.matmul(_r(p), v) / l
        ctx.save_for_backward(q, k, v, _r(o), m + torch.log2(l))
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, dO):
        q, k, v, ob, lse = ctx.saved_tensors
        scale = ctx.scale
        c = scale * 1.4426950408889634
        dvec = (dO * ob).sum(-1, keepdim=True)
        dp = torch.matmul(dO, v.transpose(2, 3)) - dvec
        p1 = torch.exp2(torch.matmul(q, _r(k * c).transpose(2, 3)) - lse)          # dK / dV kernel
        dV = torch.matmul(_r(p1).transpose(2, 3), dO)
        dK = torch.matmul(_r(p1 * dp).transpose(2, 3), q) * scale
        del p1
        p2 = torch.exp2(torch.matmul(_r(q * c), k.transpose(2, 3)) - lse)          # dQ kernel
        dQ = torch.matmul(_r(p2 * dp), k) * scale
        return dQ, dK, dV, None


class _SimsStore(torch.autograd.Function):
    """S = e_hat . q_hat^T in exact f32 (class_sims_kernel: f32 MFMA); backward: d(e_hat) in f32, d(q_hat) = bf16(dS)^T . bf16(e_hat) -- the class head's
    backward hands the prompt gradient product to the bf16 dW GEMM (autograd.py: g32 / e_bf operands of `dW(bw["g32"], bw["e_bf"], ...)`)."""
    @staticmethod
    def forward(ctx, e, q):
        ctx.save_for_backward(e, q)
        return e @ q.transpose(1, 2)

    @staticmethod
    def backward(ctx, dS):
        e, q = ctx.saved_tensors
        dq = torch.matmul(_r(dS).transpose(1, 2), _r(e)).sum(0, keepdim=True)
        return dS @ q, dq


def _layer_train(x, w, pre, heads, eps):
    B, T, D = x.shape
    dh = D // heads
    lin = lambda t, n: F.linear(t, rf(w[pre + n + ".weight"]), w[pre + n + ".bias"])
    h = rb(F.layer_norm(x, (D,), w[pre + "layer_norm1.weight"], w[pre + "layer_norm1.bias"], eps))
    q = rb(lin(h, "self_attn.q_proj")).view(B, T, heads, dh).transpose(1, 2)
    k = rb(lin(h, "self_attn.k_proj")).view(B, T, heads, dh).transpose(1, 2)
    v = rb(lin(h, "self_attn.v_proj")).view(B, T, heads, dh).transpose(1, 2)
    o = _AttnStore.apply(q, k, v, dh ** -0.5)
    o = rb(o.transpose(1, 2).reshape(B, T, D))
    x = x + rb(lin(o, "self_attn.out_proj"))
    h2 = rb(F.layer_norm(x, (D,), w[pre + "layer_norm2.weight"], w[pre + "layer_norm2.bias"], eps))
    g = rf(_ActStore.apply(lin(h2, "mlp.fc1"), "quick"))
    return x + rb(lin(g, "mlp.fc2"))


def model_forward_bf16_storage_train(cfg, w, image):
    """(boxes, sims) with the HIP path's forward AND backward rounding points; call torch.autograd.backward on them for the emulated gradients."""
    D, eps, g = cfg.hidden, cfg.ln_eps, cfg.grid
    B = image.shape[0]
    pe = F.conv2d(_r(image), _r(w["backbone.embeddings.patch_embedding.weight"]), stride=cfg.patch_size).flatten(2).transpose(1, 2)
    cls = w["backbone.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + w["backbone.embeddings.position_embedding.weight"].unsqueeze(0)
    x = F.layer_norm(x, (D,), w["backbone.pre_layernorm.weight"], w["backbone.pre_layernorm.bias"], eps)
    for i in range(cfg.layers):
        pre = f"backbone.encoder.layers.{i}."
        if any(t.requires_grad for n, t in w.items() if n.startswith(pre)) or x.requires_grad:
            x = _layer_train(x, w, pre, cfg.heads, eps)
        else:
            with torch.no_grad():
                x = _layer(x, w, pre, cfg.heads, eps)
    x = F.layer_norm(x, (D,), w["backbone.post_layernorm.weight"], w["backbone.post_layernorm.bias"], eps)
    x = x[:, 1:, :] * x[:, :1, :]
    feats = rf(F.layer_norm(x, (D,), w["post_post_layernorm.weight"], w["post_post_layernorm.bias"], eps))
    b = rf(_ActStore.apply(F.linear(feats, rf(w["box_head.dense0.weight"]), w["box_head.dense0.bias"]), "erf"))
    b = rf(_ActStore.apply(F.linear(b, rf(w["box_head.dense1.weight"]), w["box_head.dense1.bias"]), "erf"))
    b = F.linear(b, w["box_head.dense2.weight"], w["box_head.dense2.bias"])
    b = torch.sigmoid(b + O.box_bias(g).to(b.dtype))
    cx, cy, bw_, bh = b.unbind(-1)
    boxes = torch.stack([cx - 0.5 * bw_, cy - 0.5 * bh, cx + 0.5 * bw_, cy + 0.5 * bh], dim=-1)
    e = rg(F.linear(feats, rf(w["class_predictor.dense0.weight"]), w["class_predictor.dense0.bias"]))
    e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
    q = w["queries"] / torch.linalg.norm(w["queries"], dim=-1, keepdim=True) + 1e-6
    sims = F.max_pool1d(_SimsStore.apply(e, q), kernel_size=3, stride=3)
    return boxes, sims
