"""OWL-ViT vision path on MI355X -- host-side mirror of reference src/models.py.

Call surface kept from the reference:
    model = load_model(labelmap, device)            # ref src/models.py:149
    pred_boxes, None, pred_sims, None = model(image) # ref src/models.py:98-119
    model.parameters() / .train() / .eval() / named_parameters() with the reference's names, so the
    freeze rule (ref src/models.py:173-184) and `torch.optim.AdamW(model.parameters(), ...)`
    (ref main.py:56-60) work unchanged.

Everything below that surface is new: the forward and backward are sequences of hand-written HIP
kernels (libowlhip.so, C ABI in include/owl_hip.h) driven from one coarse autograd.Function; PyTorch
only owns memory, streams and the autograd edge.  Activations are bf16 with an f32 residual stream;
trainable parameters are f32 views into ONE flat bucket (and their grads into one flat grad bucket,
which is what the single RCCL all-reduce per step operates on -- SURVEY.md section 8e).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops, weights as W
from .config import OwlConfig, get_config
from .postprocess import PostProcess  # noqa: F401  (the reference exports it from src/models.py:122)


def box_bias_table(grid: int) -> torch.Tensor:
    """HF5:1071-1104 `compute_box_bias` for a square grid (computed once on the host, as HF keeps
    it as a buffer): [P,4] f32, p = y*W + x."""
    coords = torch.arange(1, grid + 1, dtype=torch.float32)
    xx, yy = torch.meshgrid(coords, coords, indexing="xy")
    bc = torch.stack((xx, yy), dim=-1)
    bc[..., 0] /= grid
    bc[..., 1] /= grid
    bc = torch.clip(bc.view(-1, 2), 0.0, 1.0)
    coord_bias = torch.log(bc + 1e-4) - torch.log1p(-bc + 1e-4)
    size = torch.full_like(coord_bias, 1.0)
    size[..., 0] /= grid
    size[..., 1] /= grid
    size_bias = torch.log(size + 1e-4) - torch.log1p(-size + 1e-4)
    return torch.cat([coord_bias, size_bias], dim=-1).contiguous()


class _Node(nn.Module):
    """Bare container so that parameter names match the reference module tree."""


def _flat_order(cfg: OwlConfig):
    """Order of the trainable tensors inside the flat bucket (q,k,v adjacent -> fused views)."""
    tl = f"backbone.encoder.layers.{cfg.trainable_layer()}."
    names = ["queries"]
    names += [tl + f"self_attn.{p}_proj.weight" for p in "qkv"] + [tl + f"self_attn.{p}_proj.bias" for p in "qkv"]
    names += [tl + "self_attn.out_proj.weight", tl + "self_attn.out_proj.bias", tl + "layer_norm1.weight", tl + "layer_norm1.bias",
              tl + "mlp.fc1.weight", tl + "mlp.fc1.bias", tl + "mlp.fc2.weight", tl + "mlp.fc2.bias",
              tl + "layer_norm2.weight", tl + "layer_norm2.bias"]
    names += ["backbone.post_layernorm.weight", "backbone.post_layernorm.bias", "post_post_layernorm.weight",
              "post_post_layernorm.bias", "class_predictor.dense0.weight", "class_predictor.dense0.bias",
              "box_head.dense0.weight", "box_head.dense0.bias", "box_head.dense1.weight", "box_head.dense1.bias",
              "box_head.dense2.weight", "box_head.dense2.bias"]
    return names


class OwlViT(nn.Module):
    """Vision-only OWL-ViT with a learnable query bank (ref src/models.py:41-119)."""

    def __init__(self, cfg: OwlConfig, state: "dict[str, np.ndarray]", device="cuda", encoder_streams: int = 2):
        super().__init__()
        self.cfg = cfg
        self.device_ = torch.device(device)
        if cfg.head_dim != 64:
            raise ValueError("attention kernels are built for head_dim = 64")
        if self.device_.type == "cuda" and torch.cuda.is_available():
            cus = torch.cuda.get_device_properties(self.device_).multi_processor_count
            if cus != ops.CHIP_CUS:
                import warnings
                warnings.warn(f"OwlViT: {self.device_} reports {cus} compute units; the kernels' grids and tile rules are laid out for the {ops.CHIP_CUS} CUs of an "
                              "unpartitioned MI355X (results stay correct, the schedule is not the measured one)")
        shapes = W.param_shapes(cfg)
        missing = set(shapes) - set(state)
        if missing:
            raise KeyError(f"missing parameters: {sorted(missing)[:4]} ...")
        order = _flat_order(cfg)
        assert set(order) == {n for n in shapes if W.is_trainable(n)}, "flat order must cover the trainable set"

        # ---- flat trainable bucket (f32) + grad bucket; every tensor starts 8-element aligned ----
        offs, off = OrderedDict(), 0
        for n in order:
            offs[n] = off
            off += (int(np.prod(shapes[n])) + 7) // 8 * 8
        self.flat_numel = off
        self.flat_offsets = offs
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=self.device_)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device_)
        self.flat_bf16 = torch.zeros(off, dtype=torch.bfloat16, device=self.device_)

        # ---- parameter tree with the reference's names ---------------------------------------------
        self._pnames = []
        for name, shape in shapes.items():
            src = torch.as_tensor(np.asarray(state[name]), dtype=torch.float32)
            assert tuple(src.shape) == tuple(shape), name
            if name in offs:
                view = self.flat_param[offs[name]: offs[name] + src.numel()].view(shape)
                view.copy_(src)
                p = nn.Parameter(view, requires_grad=True)
            else:
                p = nn.Parameter(src.to(self.device_), requires_grad=False)
            self._attach(name, p)
            self._pnames.append(name)
        self._byname = dict(self.named_parameters())
        assert list(self._byname.keys()) == list(shapes.keys()) or set(self._byname) == set(shapes)

        # ---- frozen bf16 compute copies ---------------------------------------------------------------
        D = cfg.hidden
        self._fz = {}
        P_ = self._byname
        with torch.no_grad():
            # im2row-free for every patch size (L/14's 14-pixel rows included): the weight goes into the gather loader's K order once (weights.py)
            wpe = W.patch_weight_gather_layout(P_["backbone.embeddings.patch_embedding.weight"].detach(), cfg.patch_size)
            self._fz["w_pe"] = wpe.to(torch.bfloat16).contiguous()
            for i in range(cfg.layers):
                if i == cfg.trainable_layer():
                    continue
                pre = f"backbone.encoder.layers.{i}."
                self._fz[f"{i}.wqkv"] = torch.cat([P_[pre + f"self_attn.{p}_proj.weight"] for p in "qkv"], 0).to(torch.bfloat16).contiguous()
                self._fz[f"{i}.bqkv"] = torch.cat([P_[pre + f"self_attn.{p}_proj.bias"] for p in "qkv"], 0).contiguous()
                self._fz[f"{i}.wo"] = P_[pre + "self_attn.out_proj.weight"].to(torch.bfloat16).contiguous()
                self._fz[f"{i}.w1"] = P_[pre + "mlp.fc1.weight"].to(torch.bfloat16).contiguous()
                self._fz[f"{i}.w2"] = P_[pre + "mlp.fc2.weight"].to(torch.bfloat16).contiguous()
                if i > cfg.trainable_layer():
                    # frozen layers ABOVE the trainable one (the literal "layers.11" rule on a deeper model, ref
                    # src/models.py:175): the backward passes through them (dX only) -> static transposed copies
                    for k in ("wqkv", "wo", "w1", "w2"):
                        self._fz[f"{i}.{k}T"] = self._fz[f"{i}.{k}"].t().contiguous()
        self.box_bias = box_bias_table(cfg.grid).to(self.device_)
        self._ws = {}
        self._gen = 0                      # generation of the latest forward (any batch size): see _forward_impl / autograd.OwlViTFunction
        self._ws_lru = []                  # batch sizes, most recent first: workspaces of all but the newest `max_cached_batch_sizes` are dropped
        self.max_cached_batch_sizes = 2    # (a DataLoader's ragged last batch + the regular one; every further size would pin its own activations)
        self._saved = None
        # The bf16 compute copy of the trainable bucket is re-cast by EVERY forward -- one pass over 8.7 M elements -- unless the fused AdamW has
        # just written it in its own pass: FusedAdamW.step leaves a ONE-SHOT token (this flag + the bucket's version counter) that the next
        # forward consumes.  Nothing else can set it, so writes the version counter does not see (`p.data.mul_()`, raw-pointer kernels) are picked
        # up by the next forward at the latest one forward after an optimizer step (see refresh_compute_weights for the remaining window).
        self._bf16_current = False
        self._bf16_version = None          # flat_param._version at the fused step that set the token
        self.encoder_streams = int(encoder_streams)     # sub-batches of the encoder forward, one HIP stream each (see _forward_impl); 1 = off
        self._streams, self._join, self._fork_ev, self._fork_ev_tail = [], {}, None, None
        self.head_streams = True          # box head / class head (forward and backward) on two streams when the sub-batch streams are on
        self._dw_events_ = None
        self._param_event = None           # the deferred tail of the previous step (backward / all-reduce / AdamW on the tail stream): see overlap_tail
        self._grad_clean = False           # ... which also left flat_grad zeroed for this step
        # overlap_tail (optim.FusedAdamW(..., overlap=True) / ddp.DataParallel(overlap=True) switch it on): the hand-written backward, the gradient
        # all-reduce and the fused AdamW of step i run on ONE side stream ("tail stream") while the compute stream already runs the forward of step
        # i+1 -- embeddings and encoder layers below the trainable one are frozen (ref src/models.py:173-184), so nothing before the trainable layer
        # depends on the tail; the compute stream waits for it exactly where the first trainable tensor / kept activation is touched (_wait_params).
        # The tail's small kernels (loss backward, head backward, reductions, transposes: ~1 ms at 32 workgroups or fewer) then run beside the next
        # forward's GEMMs instead of alone.  Same kernels, same operands, same order per buffer: bitwise the in-line schedule.
        self.overlap_tail = False
        self._tail_stream_ = None
        # Round 6 experiment, OFF by default: with `pretranspose` the backward's seven weight transposes (bf16 W^T of the trainable weights, the "W" operand of the
        # dX GEMMs; they do not depend on the loss) are launched by a gradient-recording forward on a stream of their own where it first reads the trainable
        # weights, and run beside the trainable layer / heads / loss chain instead of inside the backward.  Same kernels, same operands: same bits
        # (tests/test_determinism_gpu.py).  Measured (profiles/r06_tail.md): 74 us of transposes leave the backward's stream and the step does not get
        # faster (three alternated rounds: +0.17 / -0.38 / -0.23 %): the in-backward form stays the default.
        self.pretranspose = False
        self._wt, self._wt_stream_, self._wt_event = None, None, None
        self._trainable = frozenset(order)
        # checkpointing while a deferred optimizer step (ddp.DataParallel(overlap=True)) is still running on its side stream: order the
        # current stream behind it before any parameter is read
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._wait_params())

    # -- module-tree plumbing -----------------------------------------------------------------------
    def _attach(self, dotted: str, p: nn.Parameter):
        parts = dotted.split(".")
        node = self
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        node.register_parameter(parts[-1], p)

    def p(self, name: str) -> torch.Tensor:
        return self._byname[name]

    def _tview(self, name: str, dtype=torch.bfloat16):
        """bf16 (or f32) view of a trainable tensor inside the flat buckets."""
        o = self.flat_offsets[name]
        n = self._byname[name].numel()
        src = self.flat_bf16 if dtype == torch.bfloat16 else self.flat_param
        return src[o: o + n].view(self._byname[name].shape)

    def _layer_weights(self, i: int):
        cfg, D = self.cfg, self.cfg.hidden
        pre = f"backbone.encoder.layers.{i}."
        P_ = self._byname
        if i == cfg.trainable_layer():
            o = self.flat_offsets[pre + "self_attn.q_proj.weight"]
            wqkv = self.flat_bf16[o: o + 3 * D * D].view(3 * D, D)
            ob = self.flat_offsets[pre + "self_attn.q_proj.bias"]
            bqkv = self.flat_param[ob: ob + 3 * D]
            wo, w1, w2 = (self._tview(pre + "self_attn.out_proj.weight"), self._tview(pre + "mlp.fc1.weight"),
                          self._tview(pre + "mlp.fc2.weight"))
        else:
            wqkv, bqkv = self._fz[f"{i}.wqkv"], self._fz[f"{i}.bqkv"]
            wo, w1, w2 = self._fz[f"{i}.wo"], self._fz[f"{i}.w1"], self._fz[f"{i}.w2"]
        return dict(wqkv=wqkv, bqkv=bqkv, wo=wo, bo=P_[pre + "self_attn.out_proj.bias"], w1=w1, b1=P_[pre + "mlp.fc1.bias"],
                    w2=w2, b2=P_[pre + "mlp.fc2.bias"], g1=P_[pre + "layer_norm1.weight"], be1=P_[pre + "layer_norm1.bias"],
                    g2=P_[pre + "layer_norm2.weight"], be2=P_[pre + "layer_norm2.bias"])

    # -- workspaces -----------------------------------------------------------------------------------
    def _touch_batch(self, B: int):
        """Workspaces are cached per batch size; only the `max_cached_batch_sizes` most recently used sizes are kept (a third size evicts the
        least recently used one's activations, saved layers and backward scratch -- the caching allocator reuses the memory).  A backward
        still pending for an evicted size finds a fresh workspace whose generation does not match and raises (autograd.OwlViTFunction)."""
        lru = self._ws_lru
        if lru and lru[0] == B:
            return
        if B in lru:
            lru.remove(B)
        lru.insert(0, B)
        # (with a deferred tail -- overlap_tail -- the backward / optimizer of the previous step may still be reading an evicted size's buffers
        #  on the tail stream, and the caching allocator hands freed memory to the NEXT allocation on the compute stream without waiting for
        #  other streams: order this stream behind the tail before anything is dropped)
        if len(lru) > max(1, int(self.max_cached_batch_sizes)) and self._tail_stream_ is not None:
            self._wait_params()
            torch.cuda.current_stream().wait_stream(self._tail_stream_)
        while len(lru) > max(1, int(self.max_cached_batch_sizes)):
            old = lru.pop()
            for k in [k for k in self._ws if (k if isinstance(k, int) else k[1]) == old]:
                del self._ws[k]

    def _patch_scratch(self, B):
        """im2row scratch of owl_patch_embed_bf16, sized by the library's own query (the dispatcher's rule lives in csrc/gemm.hip only): a patch size that is
        not 2^n on a problem too small for the ping-pong kernel needs one; L/14 at any batch size does not."""
        cfg = self.cfg
        nbytes = torch.zeros(1, dtype=torch.int64)
        _lib.call("owl_patch_embed_scratch_bytes", B, cfg.image_size, cfg.patch_size, cfg.hidden, 0, nbytes)
        if int(nbytes.item()) == 0:
            return None
        key = ("im2row", B)
        if key not in self._ws:
            self._ws[key] = torch.zeros(int(nbytes.item()) // 2, dtype=torch.bfloat16, device=self.device_)
        return self._ws[key]

    def _workspace(self, B: int, train: bool = True):
        """Activation workspace of batch size B.  Gradient-recording forwards and no-grad (eval) forwards use SEPARATE sets, so an
        eval forward between a training forward and its backward cannot overwrite what that backward reads; two recording forwards
        at the same batch size do share one set -- the autograd node checks `gen` and refuses to run on overwritten activations."""
        self._touch_batch(B)
        key = B if train else ("eval", B)
        if key in self._ws:
            return self._ws[key]
        cfg, dev = self.cfg, self.device_
        D, I, Tp, P, Dt, C = cfg.hidden, cfg.mlp, cfg.tokens_padded, cfg.patches, cfg.text_dim, cfg.n_classes
        M, Mh = B * Tp, B * P
        bf, f32 = torch.bfloat16, torch.float32
        z = ops.zeros_rows
        ws = dict(
            x=z(M, D, f32, dev), x_fin=z(M, D, f32, dev) if train else None, h=z(M, D, bf, dev), qkv=z(M, 3 * D, bf, dev),
            att=z(M, D, bf, dev), g=z(M, I, bf, dev), d1=z(M, D, bf, dev), d2=z(M, D, bf, dev),
            # heads
            cls_ln=torch.zeros(B, D, device=dev), feats=z(Mh, D, bf, dev), st_post=torch.zeros(M, 2, device=dev),
            st_pp=torch.zeros(Mh, 2, device=dev), hb0=z(Mh, D, bf, dev), ub0=z(Mh, D, bf, dev), hb1=z(Mh, D, bf, dev),
            ub1=z(Mh, D, bf, dev), sig=torch.zeros(Mh, 4, device=dev), e=z(Mh, Dt, f32, dev),
            qhat=torch.zeros(32, Dt, device=dev), qnorm=torch.zeros(32, device=dev),
            argmax=torch.zeros(Mh, C, dtype=torch.uint8, device=dev), inv_norm=torch.zeros(Mh, device=dev),
            img=torch.zeros(B, 3, cfg.image_size, cfg.image_size, dtype=bf, device=dev),
            gen=0,          # set from the model-global counter by every recording forward (_forward_impl)
        )
        self._ws[key] = ws
        return ws

    def _layer_ws(self, B: int, i: int):
        """Saved activations of layer i (i >= trainable layer) for the backward; allocated once per batch size."""
        key = ("layer", B, i)
        if key in self._ws:
            return self._ws[key]
        cfg, dev = self.cfg, self.device_
        D, I, Tp = cfg.hidden, cfg.mlp, cfg.tokens_padded
        M = B * Tp
        bf, f32 = torch.bfloat16, torch.float32
        z = ops.zeros_rows
        L = dict(x_in=z(M, D, f32, dev), x_mid=z(M, D, f32, dev), st1=torch.zeros(M, 2, device=dev), st2=torch.zeros(M, 2, device=dev),
                 qkv=z(M, 3 * D, bf, dev), att=z(M, D, bf, dev),
                 lse=torch.zeros(B, cfg.heads, Tp, device=dev),
                 gp=z(M, I, bf, dev))          # quick_gelu'(u), saved by fc1's epilogue: what the dX GEMM through the activation multiplies by
        if i == cfg.trainable_layer():   # dW operands
            L.update(h1=z(M, D, bf, dev), h2=z(M, D, bf, dev), g=z(M, I, bf, dev))
        self._ws[key] = L
        return L

    def _wt_specs(self):
        """(name, rows, cols) of every trainable weight the backward needs transposed (autograd.backward_impl)."""
        cfg = self.cfg
        D, I, Dt = cfg.hidden, cfg.mlp, cfg.text_dim
        tl = f"backbone.encoder.layers.{cfg.trainable_layer()}."
        return [("class_predictor.dense0.weight", Dt, D), ("box_head.dense1.weight", D, D), ("box_head.dense0.weight", D, D), (tl + "mlp.fc2.weight", D, I),
                (tl + "mlp.fc1.weight", I, D), (tl + "self_attn.out_proj.weight", D, D), ("qkv", 3 * D, D)]

    def _pretranspose_weights(self):
        """Launch the backward's weight transposes now, on their own stream (called by a gradient-recording forward on the stream that has just ordered itself
        behind any deferred optimizer step, right before the trainable layer).  The backward waits for `_wt_event` and reads `_wt[name]`."""
        if self._wt is None:
            self._wt = {n: torch.empty(c, r, dtype=torch.bfloat16, device=self.device_) for n, r, c in self._wt_specs()}
            self._wt_stream_ = torch.cuda.Stream(device=self.device_)
        cur = torch.cuda.current_stream()
        s_ = self._wt_stream_
        s_.wait_stream(cur)                       # (the bf16 compute copies are current on `cur`; a previous backward's reads of these buffers precede this point on it too)
        with torch.cuda.stream(s_):
            D = self.cfg.hidden
            for n, r, c in self._wt_specs():
                if n == "qkv":
                    o = self.flat_offsets[f"backbone.encoder.layers.{self.cfg.trainable_layer()}.self_attn.q_proj.weight"]
                    src = self.flat_bf16[o: o + 3 * D * D].view(3 * D, D)
                else:
                    src = self._tview(n)
                ops.transpose_bf16(src, self._wt[n], r, c)
            ev = torch.cuda.Event()
            ev.record(s_)
        self._wt_event = ev

    def refresh_compute_weights(self, force: bool = True):
        """bf16 copies of the trainable tensors: one cast over the flat bucket.  Every forward calls it, except the first forward after a
        fused AdamW step (whose kernel writes the copy itself and leaves a one-shot token, see __init__).  The only window left: a write
        that bypasses the version counter (`p.data.copy_()`, a raw-pointer kernel) BETWEEN FusedAdamW.step() and the next forward -- call
        this method after such a write."""
        ops.cast_bf16(self.flat_param, self.flat_bf16)
        self._bf16_current = False

    def _mark_bf16_current(self):
        """optim.FusedAdamW.step only: its kernel has just rewritten flat_bf16 from the updated flat_param."""
        self._bf16_current = True
        self._bf16_version = self.flat_param._version

    def _apply(self, fn, recurse=True):
        """`.to()` / `.cuda()` / `.half()` / `.float()` / `.cpu()` replace every `param.data`, which would silently detach the 29 trainable
        tensors from the flat parameter / gradient / bf16 buckets the kernels read.  The model is built on its device in f32 (the reference's
        `.to(device)`, src/models.py:191, is done by construction): a call that changes nothing is accepted, anything else raises."""
        probe = torch.empty(0, dtype=torch.float32, device=self.device_)
        out = fn(probe)
        if out.dtype != probe.dtype or out.device != probe.device:
            raise RuntimeError(
                f"OwlViT lives on {self.device_} in float32 (trainable tensors are views of one flat bucket; compute is bf16 inside the kernels): "
                f"moving / casting the module to {out.device} / {out.dtype} is not supported -- build it with load_model(labelmap, device)")
        return self

    @property
    def _tail_stream(self):
        if self._tail_stream_ is None:
            self._tail_stream_ = torch.cuda.Stream(device=self.device_)
        return self._tail_stream_

    def finish(self):
        """Order the current stream behind a deferred tail (overlap_tail): call before reading parameters or gradients outside the model's own
        forward / backward / zero_grad / state_dict, which do it themselves."""
        self._wait_params()

    def _wait_params(self):
        """Order the compute stream behind a deferred optimizer step (ddp.DataParallel(overlap=True)): called right before the
        first trainable tensor is read, i.e. after the frozen prefix (embeddings + encoder layers below the trainable one)."""
        if self._param_event is not None:
            torch.cuda.current_stream().wait_event(self._param_event)
            self._param_event = None

    def _check_trainable_set(self):
        """The backward is built for exactly the reference's freeze rule (ref src/models.py:173-184).  The reference's rule is a
        user-editable loop over requires_grad; here a different set would silently get no gradient (or still be updated by the
        optimizer), so a mismatch is an error rather than a silent difference."""
        for n, p in self._byname.items():
            if p.requires_grad != (n in self._trainable):
                raise RuntimeError(
                    f"parameter `{n}` has requires_grad={p.requires_grad}, but this build's hand-written backward computes gradients for "
                    "exactly the reference's trainable set (layers.11 / box / post_layernorm / class_predictor / queries, ref "
                    "src/models.py:173-184); freezing or unfreezing individual tensors is not supported")

    # -- encoder schedule --------------------------------------------------------------------------------
    def _encoder_chunks(self, B: int):
        """[(first image, images)] of the sub-batches the encoder runs on separate streams.  Batches below 4 stay whole (train step, same process:
        batch 2 -7 %, batch 4 +8 %, batch 6 +0.5 %, batch 8 +4 % with two sub-batches)."""
        n = self.encoder_streams if B >= 4 else 1
        if n <= 1:
            return [(0, B)]
        base, extra = divmod(B, n)
        out, b0 = [], 0
        for c in range(n):
            nb = base + (1 if c < extra else 0)
            out.append((b0, nb)); b0 += nb
        return out

    @property
    def _dw_events(self):
        """Events of the backward's weight-gradient stream (autograd.backward_impl): four operand-ready events + the final join."""
        if self._dw_events_ is None:
            self._dw_events_ = [torch.cuda.Event() for _ in range(5)]
        return self._dw_events_

    def _side_stream(self, c: int):
        while len(self._streams) < c:
            self._streams.append(torch.cuda.Stream(device=self.device_))
            self._join[len(self._streams)] = torch.cuda.Event()
        return self._streams[c - 1]

    def _bwd_streams(self):
        """(side stream c, join event c, fork event) as the backward uses them: the forward's own in-line, a disjoint set as a deferred tail."""
        off = max(self.encoder_streams - 1, 1) if self.overlap_tail else 0
        if off and self._fork_ev_tail is None:
            self._fork_ev_tail = torch.cuda.Event()
        if not off and self._fork_ev is None:
            self._fork_ev = torch.cuda.Event()
        S = lambda c: self._side_stream(c + off)
        J = lambda c: (self._side_stream(c + off), self._join[c + off])[1]
        return S, J, (self._fork_ev_tail if off else self._fork_ev)

    def _encoder_layer(self, i: int, ws, B: int, save: bool, st):
        """Encoder layer i (HF5:478-511) for the sub-batch `st` (images [b0, b0 + nb) = rows [b0 Tp, (b0 + nb) Tp) of every buffer), on the
        current stream.  st carries the sub-batch's residual stream and the not-yet-added branch outputs from layer to layer."""
        cfg = self.cfg
        D, I, H, Tp, T = cfg.hidden, cfg.mlp, cfg.heads, cfg.tokens_padded, cfg.tokens
        b0, nb = st["b0"], st["nb"]
        r0, M = b0 * Tp, nb * Tp
        R = lambda t: t[r0:r0 + M]
        tl = cfg.trainable_layer()
        scale = cfg.head_dim ** -0.5
        xs, pending, pending1 = st["xs"], st["pending"], st["pending1"]
        d1, d2 = R(ws["d1"]), R(ws["d2"])
        lw = self._layer_weights(i)
        sv = save and i >= tl          # the backward passes through this layer: keep its activations
        full = sv and i == tl          # ... and, for the trainable layer, the dW operands too
        Ls = self._layer_ws(B, i) if sv else None
        st1 = R(Ls["st1"]) if sv else None
        h = R(Ls["h1"]) if full else R(ws["h"])
        # Residual adds (HF5:500,507) live in the LayerNorm kernels: the GEMM in front of each emits a bf16
        # delta through the fast wide-store epilogue, and LN does x += delta while it normalises.
        x_cur = R(Ls["x_in"]) if sv else xs
        if pending is None:
            if sv:
                x_cur.copy_(xs)
            ops.layernorm(x_cur, lw["g1"], lw["be1"], h, M, D, st1, cfg.ln_eps)
        else:
            if pending1 is None:
                ops.layernorm(xs, lw["g1"], lw["be1"], h, M, D, st1, cfg.ln_eps, delta=pending, x_out=x_cur)
            else:       # (xs + delta1) + delta2, same operands and order as the two separate adds
                ops.layernorm(xs, lw["g1"], lw["be1"], h, M, D, st1, cfg.ln_eps, delta=pending1, delta2=pending, x_out=x_cur)
        # ONE row-major QKV GEMM per layer.  The attention kernels (forward and backward) read every transposed MFMA operand
        # (V^T; Q^T, K^T, dO^T) out of the row-major tiles with the LDS hardware transpose (ds_read_b64_tr_b16): no transposed
        # copy of anything exists in HBM.
        qkv_l = R(Ls["qkv"]) if sv else R(ws["qkv"])
        cc = st.get("conc", 1)          # sub-batch streams in flight: small problems running ALONE take half-height GEMM tiles (ops.gemm)
        ops.gemm(ops.EPI_BIAS_BF16, h, lw["wqkv"], qkv_l, bias=lw["bqkv"], M=M, N=3 * D, K=D, ldo=3 * D, concurrency=cc)
        att_l = R(Ls["att"]) if sv else R(ws["att"])
        ops.attention_fwd_vrow(qkv_l, qkv_l[:, D:], qkv_l[:, 2 * D:], 3 * D, att_l, D, Ls["lse"][b0:b0 + nb] if sv else None, nb, H, T, Tp, scale)
        ops.gemm(ops.EPI_BIAS_BF16, att_l, lw["wo"], d1, bias=lw["bo"], M=M, N=D, K=D, concurrency=cc)
        x_mid = R(Ls["x_mid"]) if sv else x_cur
        h2 = R(Ls["h2"]) if full else R(ws["h"])
        # A frozen layer's x + delta1 is read by nobody but the next LayerNorm: it is not stored (4 bytes per element), that
        # LayerNorm adds both branch outputs instead (2 more bytes read).  Layers whose activations are kept, and the last one
        # (the merge kernel takes a single delta), store it.
        defer = (not sv) and (i + 1 < cfg.layers)
        ops.layernorm(x_cur, lw["g2"], lw["be2"], h2, M, D, R(Ls["st2"]) if sv else None, cfg.ln_eps, delta=d1, x_out=x_mid,
                      store_x=not defer)
        g_l = R(Ls["g"]) if full else R(ws["g"])
        ops.gemm(ops.EPI_QGELU_BF16, h2, lw["w1"], g_l, bias=lw["b1"], aux=R(Ls["gp"]) if sv else None, M=M, N=I, K=D, concurrency=cc)
        ops.gemm(ops.EPI_BIAS_BF16, g_l, lw["w2"], d2, bias=lw["b2"], M=M, N=D, K=I, concurrency=cc)
        st["pending"], st["xs"] = d2, x_mid
        st["pending1"] = d1 if defer else None

    # -- forward ---------------------------------------------------------------------------------------
    def _forward_impl(self, image: torch.Tensor, save: bool):
        cfg = self.cfg
        D, I, H, Tp, T, P, Dt, C = cfg.hidden, cfg.mlp, cfg.heads, cfg.tokens_padded, cfg.tokens, cfg.patches, cfg.text_dim, cfg.n_classes
        B = image.shape[0]
        if tuple(image.shape[1:]) != (3, cfg.image_size, cfg.image_size):
            raise ValueError(f"image must be [B,3,{cfg.image_size},{cfg.image_size}], got {tuple(image.shape)}")
        ws = self._workspace(B, train=save)
        # model-global, monotonically increasing: a workspace rebuilt after an LRU eviction can never carry a generation an older autograd
        # node still holds (a per-workspace counter restarted at 0 and could collide: fwd(A) -> two other sizes evict A -> fwd(A) again)
        self._gen += 1
        ws["gen"] = self._gen
        M, Mh = B * Tp, B * P
        P_ = self._byname
        if self._bf16_current and self._bf16_version == self.flat_param._version:
            self._bf16_current = False          # one-shot: the fused AdamW's own bf16 pass covers exactly this forward
        else:
            self._wait_params()
            self.refresh_compute_weights()

        if image.dtype == torch.float32:
            ops.cast_bf16(image.contiguous(), ws["img"])
            img = ws["img"]
        elif image.dtype == torch.bfloat16:
            img = image.contiguous()
        else:
            raise TypeError("image must be float32 or bfloat16")

        x = ws["x"]
        ops.patch_embed(img, self._fz["w_pe"], P_["backbone.embeddings.position_embedding.weight"], x, B, cfg.image_size,
                        cfg.patch_size, D, Tp, scratch=self._patch_scratch(B))
        ops.cls_rows(x, P_["backbone.embeddings.class_embedding"], P_["backbone.embeddings.position_embedding.weight"], B, Tp, D)
        ops.layernorm(x, P_["backbone.pre_layernorm.weight"], P_["backbone.pre_layernorm.bias"], x, M, D, eps=cfg.ln_eps)

        # ---- encoder.  No kernel of it couples images, so the batch is run as `encoder_streams` sub-batches (contiguous row ranges of the
        #      same buffers), each on its own HIP stream, layer by layer: the idle CUs of one sub-batch's last GEMM round / attention tail
        #      run the other sub-batch's next kernel (a 256 x 256 GEMM workgroup owns its CU's registers and LDS, so the overlap is at CU
        #      granularity).  Measured on the forward: -3.7 % at B/16 batch 32, -2.9 % at L/14 batch 16 (two streams; three or four lose --
        #      profiles/r02_encoder_streams.md).  Every kernel is batch-invariant: same bits as the single-stream schedule.
        tl = cfg.trainable_layer()
        param_event, self._param_event = self._param_event, None
        chunks = self._encoder_chunks(B)
        main = torch.cuda.current_stream()
        if save:        # first use allocates (and zero-fills, on THIS stream) the kept activations: before any other stream may write them
            for i in range(tl, cfg.layers):
                self._layer_ws(B, i)
        if len(chunks) > 1:
            if self._fork_ev is None:
                self._fork_ev = torch.cuda.Event()
            self._fork_ev.record(main)
        states = []
        for c, (b0, nb) in enumerate(chunks):
            st = dict(b0=b0, nb=nb, stream=main if c == 0 else self._side_stream(c), xs=x[b0 * Tp:(b0 + nb) * Tp], pending=None, pending1=None, conc=len(chunks))
            if c > 0:
                st["stream"].wait_event(self._fork_ev)
            states.append(st)
        for i in range(cfg.layers):
            for st in states:
                with torch.cuda.stream(st["stream"]):
                    if i == tl and param_event is not None:
                        st["stream"].wait_event(param_event)    # everything above ran on frozen weights only (ddp overlap schedule)
                    if i == tl and save and self.pretranspose and st is states[0]:
                        self._pretranspose_weights()            # the backward's W^T copies, beside the rest of this forward and the loss chain
                    self._encoder_layer(i, ws, B, save, st)
        for c, st in enumerate(states):
            if c > 0:
                self._join[c].record(st["stream"])
                main.wait_event(self._join[c])
        # the sub-batches' residual streams / deferred MLP-branch outputs are row ranges of ONE buffer each: the merge kernel takes the batch
        last = cfg.layers - 1
        xs = self._layer_ws(B, last)["x_mid"] if (save and last >= tl) else x
        assert all(st["xs"].data_ptr() == xs.data_ptr() + st["b0"] * Tp * D * 4 for st in states)
        pending = ws["d2"]

        # ---- final residual add + post_layernorm (all tokens) * class token -> post_post_layernorm
        #      (ref src/models.py:80-86); the final residual stream is materialised in `x` for the backward
        feats = ws["feats"]
        tv = lambda n: self._tview(n)
        # (the backward reads the final residual stream; it gets a buffer of its own because with overlap_tail the NEXT forward's patch embedding
        #  rewrites `x` while this step's backward may still be running)
        ops.merge_ln(xs, P_["backbone.post_layernorm.weight"], P_["backbone.post_layernorm.bias"], P_["post_post_layernorm.weight"],
                     P_["post_post_layernorm.bias"], ws["cls_ln"], feats, ws["st_post"], ws["st_pp"], B, P, Tp, D, cfg.ln_eps,
                     delta=pending, x_out=ws["x_fin"] if save else x)
        # The box head and the class head only share their input: with sub-batch streams on, the class head runs on the side stream beside the
        # box head (its kernels fill the CUs the box GEMMs' remainder rounds leave idle).  Outputs are allocated before the fork.
        pred_boxes = torch.empty(B, P, 4, device=self.device_)
        pred_sims = torch.empty(B, P, C, device=self.device_)
        side = self._side_stream(1) if (self.head_streams and self.encoder_streams > 1 and len(chunks) > 1) else None
        if side is not None:
            self._fork_ev.record(main)
            side.wait_event(self._fork_ev)
        # ---- class head (ref src/models.py:24-38) ------------------------------------------------------
        with torch.cuda.stream(side if side is not None else main):
            ops.gemm(ops.EPI_F32, feats, tv("class_predictor.dense0.weight"), ws["e"], bias=P_["class_predictor.dense0.bias"], M=Mh, N=Dt, K=D)
            ops.query_normalize(P_["queries"], ws["qhat"], ws["qnorm"], cfg.queries, Dt)
            ops.class_sims(ws["e"], ws["qhat"], pred_sims, ws["argmax"], ws["inv_norm"], Mh, Dt, C)
        # ---- box head (HF5:983-999) + bias / sigmoid / corners ---------------------------------------
        ops.gemm(ops.EPI_GELU_BF16, feats, tv("box_head.dense0.weight"), ws["hb0"], bias=P_["box_head.dense0.bias"],
                 aux=ws["ub0"] if save else None, M=Mh, N=D, K=D)
        ops.gemm(ops.EPI_GELU_BF16, ws["hb0"], tv("box_head.dense1.weight"), ws["hb1"], bias=P_["box_head.dense1.bias"],
                 aux=ws["ub1"] if save else None, M=Mh, N=D, K=D)
        ops.box_final(ws["hb1"], P_["box_head.dense2.weight"], P_["box_head.dense2.bias"], self.box_bias, pred_boxes, ws["sig"], Mh, P, D)
        if side is not None:
            self._join[1].record(side)
            main.wait_event(self._join[1])
        return pred_boxes, pred_sims

    def forward(self, image: torch.Tensor):
        """ref src/models.py:98-119: returns (pred_boxes xyxy, None, pred_sims, None)."""
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            self._check_trainable_set()
            from .autograd import OwlViTFunction
            names = list(self.flat_offsets.keys())
            boxes, sims = OwlViTFunction.apply(self, image, *[self._byname[n] for n in names])
        else:
            boxes, sims = self._forward_impl(image, save=False)
        return (boxes, None, sims, None)


def load_model(labelmap, device="cuda", arch: str = "owlvit-base-patch32", seed: int = 1234, state=None, *,
               prompt_ids=None, text_state=None, vocab=None, merges=None):
    """ref src/models.py:149-191.  The reference downloads `google/owlvit-base-patch32` and runs the
    CLIP text tower once to initialise the query bank; neither box has network access, so weights
    come from `state` (name -> array, reference parameter names) or, by default, the deterministic
    random set (weights.make_weights).  The freeze rule is applied by construction (trainable
    tensors live in the flat bucket with requires_grad=True; everything else is frozen).

    `prompt_ids` ([3C, S] CLIP token ids of `[label, "a photo of "+label, "a "+label+" in an environment"]`
    per class, class-major, ref models.py:155-159 -- tokenised by the caller's processor) switches on the
    reference's query-bank initialisation: the text tower (text.TextTower; weights from `text_state`, HF names
    `text_model.*` / `text_projection.weight`, or the deterministic random set) runs once on the device and its
    L2-normalised `text_embeds` become `queries` (ref models.py:161-169).

    `vocab` / `merges` (paths of the CLIP `vocab.json` / `merges.txt` every OWL-ViT checkpoint carries -- a download neither box can make, so they
    are not shipped): the three prompts per label are built and tokenised HERE exactly as the reference does (ref models.py:155-166 via
    tokenizer.ClipBPE, id for id `transformers.CLIPTokenizer`), i.e. the unchanged `load_model(labelmap, device)` call of ref main.py:42 plus the two
    paths gives the reference's query bank."""
    n_classes = len(labelmap)
    cfg = get_config(arch, n_classes=n_classes)
    if (vocab is None) != (merges is None):
        raise ValueError("load_model: vocab= and merges= go together (the CLIP vocab.json and merges.txt)")
    if vocab is not None:
        if prompt_ids is not None:
            raise ValueError("load_model: give either prompt_ids= or vocab= / merges=")
        from .config import get_text_config
        from .tokenizer import ClipBPE, label_prompts
        lm = labelmap if hasattr(labelmap, "values") else {i: l for i, l in enumerate(labelmap)}
        prompt_ids = ClipBPE(vocab, merges, max_length=get_text_config(arch).max_pos)(label_prompts(lm))
    if state is None:
        state = W.make_weights(cfg, seed)
    if prompt_ids is not None:
        from .config import get_text_config
        from .text import TextTower
        ids = np.asarray(prompt_ids.cpu() if torch.is_tensor(prompt_ids) else prompt_ids)
        if ids.ndim == 3:                         # processor(text=[to_encode]) yields [1, 3C, S]
            ids = ids[0]
        if ids.shape[0] != cfg.queries:
            raise ValueError(f"load_model: expected {cfg.queries} prompts (3 per class), got {ids.shape[0]}")
        tower = TextTower(get_text_config(arch), text_state, device, seed)
        state = OrderedDict(state)
        state["queries"] = tower.query_bank(ids).cpu().numpy()
        del tower
    return OwlViT(cfg, state, device)
