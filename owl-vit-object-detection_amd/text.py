"""Query-bank initialisation on device (ref src/models.py:155-169).

The reference runs HF's CLIP-style text tower once over ``[label, "a photo of "+label, "a "+label+" in an
environment"]`` per class and keeps ``text_embeds`` (L2-normalised, [1, 3C, Dt]) as the learnable ``queries``.
``TextTower.encode(input_ids)`` reproduces ``_model(**inputs).text_embeds`` for already-tokenised prompts (the CLIP BPE
vocabulary is a download the boxes cannot make, so tokenisation stays with the caller's ``processor(text=...)``).

Pre-LN causal transformer (HF5:603-663): token+position embedding -> L x [LN1 -> QKV GEMM -> causal attention ->
out-proj (+residual) -> LN2 -> fc1+quick-GELU -> fc2 (+residual)] -> final LN of the EOS row -> text_projection ->
L2 normalise.  Linear/LayerNorm layers are the vision path's HIP kernels (bf16 MFMA, f32 residual stream); the
embedding gather, the short causal attention and the pool/projection tail are csrc/text.hip.  Padding keys need no
mask: they sit after the EOS token, which the causal mask already hides from the pooled row.
"""
import torch

from . import _lib, ops
from . import weights as W_
from .config import TextConfig


class TextTower:
    def __init__(self, tcfg: TextConfig, state=None, device="cuda", seed: int = 1234):
        self.cfg = tcfg
        self.device = torch.device(device)
        if tcfg.width % 64 or tcfg.width // tcfg.heads != 64:
            raise ValueError("TextTower: head dim must be 64")
        src = state if state is not None else W_.make_text_weights(tcfg, seed)
        g = {}
        for name, shape in W_.text_param_shapes(tcfg).items():
            key = name if name in src else "owlvit." + name
            if key not in src:
                raise KeyError(f"TextTower: missing parameter {name}")
            t = torch.as_tensor(src[key]).detach().to(torch.float32).reshape(shape).contiguous()
            g[name] = t.to(self.device)
        self.p = g
        bf = torch.bfloat16
        self.layers = []
        for i in range(tcfg.layers):
            pre = f"text_model.encoder.layers.{i}."
            self.layers.append(dict(
                wqkv=torch.cat([g[pre + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).to(bf).contiguous(),
                bqkv=torch.cat([g[pre + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0).contiguous(),
                wo=g[pre + "self_attn.out_proj.weight"].to(bf).contiguous(), bo=g[pre + "self_attn.out_proj.bias"],
                w1=g[pre + "mlp.fc1.weight"].to(bf).contiguous(), b1=g[pre + "mlp.fc1.bias"],
                w2=g[pre + "mlp.fc2.weight"].to(bf).contiguous(), b2=g[pre + "mlp.fc2.bias"],
                g1=g[pre + "layer_norm1.weight"], be1=g[pre + "layer_norm1.bias"],
                g2=g[pre + "layer_norm2.weight"], be2=g[pre + "layer_norm2.bias"]))

    @torch.no_grad()
    def encode(self, input_ids) -> torch.Tensor:
        """input_ids [N, S] int64 (S <= max_pos) -> text_embeds [N, proj_dim] f32, rows L2-normalised (HF5:945-970)."""
        c = self.cfg
        ids = torch.as_tensor(input_ids).to(self.device, torch.int64).contiguous()
        if ids.dim() != 2 or ids.shape[1] > c.max_pos:
            raise ValueError(f"TextTower.encode: expected [N, S<={c.max_pos}] token ids, got {tuple(ids.shape)}")
        N, S = ids.shape
        Wd, M = c.width, N * S
        dev, bf, f32 = self.device, torch.bfloat16, torch.float32
        x = ops.zeros_rows(M, Wd, f32, dev)
        h = ops.zeros_rows(M, Wd, bf, dev)
        qkv = ops.zeros_rows(M, 3 * Wd, bf, dev)
        att = ops.zeros_rows(M, Wd, bf, dev)
        u = ops.zeros_rows(M, c.mlp, bf, dev)
        st = ops.stream()
        _lib.call("owl_text_embed", st, ids, self.p["text_model.embeddings.token_embedding.weight"],
                  self.p["text_model.embeddings.position_embedding.weight"], x, N, S, Wd, c.vocab)
        for lw in self.layers:
            ops.layernorm(x, lw["g1"], lw["be1"], h, M, Wd, eps=c.ln_eps)
            ops.gemm(ops.EPI_BIAS_BF16, h, lw["wqkv"], qkv, bias=lw["bqkv"], M=M, N=3 * Wd, K=Wd)
            _lib.call("owl_causal_attention_small", st, qkv, att, N, S, c.heads, 0.125)
            ops.gemm(ops.EPI_RESID_F32, att, lw["wo"], x, bias=lw["bo"], resid=x, M=M, N=Wd, K=Wd)
            ops.layernorm(x, lw["g2"], lw["be2"], h, M, Wd, eps=c.ln_eps)
            ops.gemm(ops.EPI_QGELU_BF16, h, lw["w1"], u, bias=lw["b1"], M=M, N=c.mlp, K=Wd)
            ops.gemm(ops.EPI_RESID_F32, u, lw["w2"], x, bias=lw["b2"], resid=x, M=M, N=Wd, K=c.mlp)
        out = torch.empty(N, c.proj_dim, dtype=f32, device=dev)
        _lib.call("owl_text_pool_project", st, x, ids, self.p["text_model.final_layer_norm.weight"],
                  self.p["text_model.final_layer_norm.bias"], self.p["text_projection.weight"], out, N, S, Wd, c.proj_dim,
                  float(c.ln_eps))
        return out

    def query_bank(self, input_ids) -> torch.Tensor:
        """-> [1, N, proj_dim]: the ``query_bank`` argument of the reference's OwlViT wrapper (ref models.py:166-169)."""
        return self.encode(input_ids)[None]
