"""The autograd edge of the HIP train path: ONE coarse torch.autograd.Function around the whole vision
model.  forward = OwlViT._forward_impl (saving the trainable layer's activations); backward = a fixed
sequence of hand-written HIP kernels (csrc/backward.hip, attention_bwd.hip, gemm.hip) for exactly the
trainable set of the reference freeze rule (ref src/models.py:173-184): queries, encoder layer 11,
post_layernorm, post_post_layernorm, class_predictor.dense0, box_head.  Everything upstream of the
trainable layer is frozen, so the backward stops there (ref main.py:90).

Parameter gradients are ACCUMULATED by the kernels directly into `model.flat_grad` (each parameter's
.grad is a view of that bucket), so `loss.backward()` leaves one contiguous buffer ready for the single
all-reduce + optimizer step; the Function therefore returns no per-parameter tensors.
"""
import torch

from . import _lib, ops


DW_ITEMS = 256             # (tile, split) work items a weight-gradient GEMM is cut into: one per CU (bench.py --dw-items: fewer = less slab traffic and CUs left to the dX chain; measured, profiles/r06_tail.md)
TN_SMALL_N = True          # round 6: dW products with fewer than 256 output rows (the 32 x Dt prompt gradient) on the TN kernel too; False = explicit transposes + NT split-K (A/B)
FOLD_BIAS_COLSUM = True    # round 6: bias gradients of the TN-kernel Linears come out of the dW GEMM's own pass (csrc/gemm_tn.hip); False = the separate colsum_bf16 pass (A/B: bench.py --fold-bias 0)


def _grads_attached(model) -> bool:
    """Every trainable parameter's .grad is its view of model.flat_grad."""
    for n, off in model.flat_offsets.items():
        g = model._byname[n].grad
        if g is None or g.data_ptr() != model.flat_grad.data_ptr() + 4 * off:
            return False
    return True


def _attach_grads(model):
    """Make every trainable parameter's .grad a view of model.flat_grad.  If an optimizer set them to
    None (zero_grad(set_to_none=True)) the bucket is zeroed first -- None means zero.  (Callers with a deferred tail order the
    current stream behind it first: optim.FusedAdamW._attach.)"""
    if _grads_attached(model):
        return
    keep = {}
    for n in model.flat_offsets:
        g = model._byname[n].grad
        if g is not None:
            keep[n] = g
    model.flat_grad.zero_()
    for n, off in model.flat_offsets.items():
        p = model._byname[n]
        view = model.flat_grad[off: off + p.numel()].view(p.shape)
        if n in keep and keep[n].data_ptr() != view.data_ptr():
            view.copy_(keep[n])          # foreign accumulated grads are preserved
        p.grad = view


class OwlViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, image, *params):
        boxes, sims = model._forward_impl(image, save=True)
        ctx.model = model
        ctx.B = image.shape[0]
        ctx.sims = sims
        ctx.gen = model._workspace(ctx.B)["gen"]
        return boxes, sims

    @staticmethod
    def backward(ctx, d_boxes, d_sims):
        model, B = ctx.model, ctx.B
        if model._workspace(B)["gen"] != ctx.gen:
            # saved activations live in per-batch-size workspaces (not in ctx): a later gradient-recording forward at this batch
            # size has overwritten them (e.g. two forwards, then (l1 + l2).backward())
            raise RuntimeError(
                f"OwlViT backward: the activations of this forward (batch size {B}) were overwritten by a later gradient-recording "
                "forward at the same batch size (or evicted: only the model's `max_cached_batch_sizes` most recent batch sizes keep their "
                "workspaces); call backward() before the next training forward (no-grad / eval forwards are fine)")
        if getattr(model, "overlap_tail", False) and d_boxes.is_cuda:
            # deferred tail (models.OwlViT.overlap_tail): the whole backward goes to the tail stream; the compute stream carries on with whatever
            # the caller enqueues next (the next forward's frozen prefix) and waits for the event where it touches a trainable tensor
            main, ts = torch.cuda.current_stream(), model._tail_stream
            ts.wait_stream(main)
            for t in (d_boxes, d_sims, ctx.sims):
                t.record_stream(ts)               # (allocated on the compute stream, read on the tail stream)
            with torch.cuda.stream(ts):
                backward_impl(model, B, d_boxes, d_sims, ctx.sims)
                ev = torch.cuda.Event()
                ev.record(ts)
            model._param_event = ev
        else:
            backward_impl(model, B, d_boxes, d_sims, ctx.sims)
        return (None, None) + (None,) * len(model.flat_offsets)


def _bws(model, B):
    key = ("bwd", B)
    if key in model._ws:
        return model._ws[key]
    cfg, dev = model.cfg, model.device_
    D, I, Tp, P, Dt, H = cfg.hidden, cfg.mlp, cfg.tokens_padded, cfg.patches, cfg.text_dim, cfg.heads
    M, Mh = B * Tp, B * P
    Mp, Mhp = ops.pad_rows(M), ops.pad_rows(Mh)
    bf, f32 = torch.bfloat16, torch.float32
    z = ops.zeros_rows
    wide = max(3 * D, I)
    tn_all = all(v % 256 == 0 for v in (D, I, Dt))          # every token-row dW goes through the TN kernel (gemm_tn.hip)
    ws = dict(
        de=z(Mh, Dt, bf, dev), dqhat=torch.zeros(32, Dt, device=dev), du1=z(Mh, D, bf, dev), du0=z(Mh, D, bf, dev),
        g32=z(Mh, 32, bf, dev), e_bf=z(Mh, Dt, bf, dev),
        box_part=torch.zeros(_lib.load().owl_box_final_bwd_blocks(Mh), 5 * D + 4, device=dev),
        slab=torch.zeros(_slab_elems(cfg), device=dev),
        # per-split partial sums of the bias gradients (one row of n_out floats per split of the dW GEMM; at most 256 splits); the class head's chain has its own
        bslab=torch.zeros(256 * max(3 * D, I, Dt), device=dev), bslab2=torch.zeros(256 * max(D, Dt), device=dev),
        dfeats=z(Mh, D, f32, dev), dcls=torch.zeros(B, D, device=dev),
        dx=z(M, D, f32, dev), dxb=z(M, D, bf, dev), du=z(M, I, bf, dev), dh=z(M, D, bf, dev), dxm=z(M, D, f32, dev),
        datt=z(M, D, bf, dev), dqkv=z(M, 3 * D, bf, dev),
        dvec=torch.zeros(B, H, Tp, device=dev),
        # partial sums of the deterministic row reductions (bias / LayerNorm-affine gradients): written by one kernel, added in a
        # fixed order by the next -- no f32 atomics into the gradient bucket
        part=ops.rowreduce_workspace(B, Tp, max(3 * D, I, Dt), dev),
        part2=ops.rowreduce_workspace(B, Tp, max(3 * D, I, Dt), dev),      # ... of the weight-gradient stream (see backward_impl)
        dxb2=z(M, D, bf, dev),                                              # second bf16 dx of the trainable layer (the first one is still being read)
        # transposed-operand scratch for the dW GEMMs; token-row and head-row users get their own buffers so
        # that the zero pad columns [rows, rows_pad) of each are never dirtied by the other row count
        # (only the shapes the TN kernel does not take need them: feature counts that are not multiples of 256 -- the
        # parity-test configs -- and the 32 x Dt prompt-gradient product)
        tA=None if tn_all else torch.zeros(wide, Mp, dtype=bf, device=dev),
        tB=None if tn_all else torch.zeros(wide, Mp, dtype=bf, device=dev),
        tAh=torch.zeros(32 if tn_all else max(D, Dt), Mhp, dtype=bf, device=dev),
        tBh=torch.zeros(max(D, Dt), Mhp, dtype=bf, device=dev),
        wT=torch.zeros(wide * max(D, I), dtype=bf, device=dev),
        # the class head's backward runs beside the box head's on the side stream: its own split-K slab and transposed-weight scratch
        slab2=torch.zeros(max(_split_k(a, b, 1 << 30) * a * b for a, b in ((Dt, D), (32, Dt))), device=dev),
        wT2=torch.zeros(Dt * D, dtype=bf, device=dev),
        tn_all=tn_all,
    )
    model._ws[key] = ws
    return ws


def _split_k(n_rows_out, n_cols_out, k):
    """Split the (long) token contraction so that a dW GEMM gives every CU about one work item."""
    t = 256 if (n_rows_out >= 512 and n_cols_out >= 256) else 128          # tile the C side picks (gemm.hip `launch`)
    tiles = ((n_rows_out + t - 1) // t) * ((n_cols_out + t - 1) // t)
    slots = 256 if t == 256 else 512
    return max(1, min(k // 64, slots // tiles))


def _slab_elems(cfg):
    """f32 elements of the split-K slab scratch: max over the dW shapes of splits * n_out * n_in."""
    D, I, Dt = cfg.hidden, cfg.mlp, cfg.text_dim
    shapes = [(3 * D, D), (D, D), (I, D), (D, I), (Dt, D), (32, Dt)]
    return max(_split_k(a, b, 1 << 30) * a * b for a, b in shapes)


def backward_impl(model, B, d_boxes, d_sims, sims):
    cfg = model.cfg
    D, I, H, Tp, T, P, Dt, C = cfg.hidden, cfg.mlp, cfg.heads, cfg.tokens_padded, cfg.tokens, cfg.patches, cfg.text_dim, cfg.n_classes
    M, Mh = B * Tp, B * P
    Mp, Mhp = ops.pad_rows(M), ops.pad_rows(Mh)
    ws, bw = model._workspace(B), _bws(model, B)
    P_ = model._byname
    model._wait_params()
    model._grad_clean = False
    _attach_grads(model)
    G = lambda n: P_[n].grad                      # views into model.flat_grad (accumulated into)
    tv = model._tview
    tl = f"backbone.encoder.layers.{cfg.trainable_layer()}."
    d_boxes = d_boxes.contiguous().float()
    d_sims = d_sims.contiguous().float()

    # the forward launched the transposes on their own stream (models.OwlViT._pretranspose_weights): order this stream behind them once
    pre_wt = model._wt if (getattr(model, "pretranspose", False) and model._wt_event is not None) else None
    if pre_wt is not None:
        torch.cuda.current_stream().wait_event(model._wt_event)

    def wT(name, rows, cols, buf="wT"):
        """bf16 transpose of a trainable weight [rows, cols] -> [cols, rows]: the forward's pre-transposed copy, or (pretranspose off) made here in scratch."""
        if pre_wt is not None:
            return pre_wt[name]
        out = bw[buf][: rows * cols].view(cols, rows)
        ops.transpose_bf16(tv(name), out, rows, cols)
        return out

    def dW(dy, x, grad_w, n_out, n_in, rows, rows_pad, grad_b=None, accumulate=1, part="part", slab="slab"):
        """grad_w[n_out, n_in] (+)= dy[rows, n_out]^T x[rows, n_in];  grad_b += colsum(dy).
        Token-major operand copies (transposes) -> split-K GEMM into f32 slabs -> deterministic slab reduction.
        Pad columns [rows, rows_pad) of the scratch stay zero: never written, buffers start zeroed."""
        # (round 6: also the class head's 32 x Dt prompt-gradient product -- one 256-wide n tile of which 32 rows are kept; it used to go through two explicit
        #  transposes + the NT split-K kernel: 92 us against ~30, bench.py --tn-small-n 0 / 1, profiles/r06_tail.md)
        if n_in % 256 == 0 and (n_out % 256 == 0 or (TN_SMALL_N and n_out % 8 == 0 and n_out <= 64 and bw["tn_all"] and grad_b is None)):
            # TN kernel: reads dy / x where they lie (LDS transpose-reads), no token-major copies.  The bias gradient (column sums of dy) comes out of the
            # same pass as per-split partial sums (round 6: it was a second read of dy by a kernel of its own) and is added up like the weight slabs
            tiles = ((n_out + 255) // 256) * (n_in // 256)
            bs = bw["bslab2" if slab == "slab2" else "bslab"] if (grad_b is not None and FOLD_BIAS_COLSUM) else None
            if grad_b is not None and bs is None:          # (A/B switch off: the column-sum kernel of rounds 2-5)
                ops.colsum_bf16(dy, grad_b, rows, n_out, partials=bw[part])
            ns = ops.gemm_tn_slab(dy, x, bw[slab], rows, n_out, n_in, max(1, min(DW_ITEMS, 256) // tiles), bias_slab=bs)          # (the slab scratch is sized for 256 items)
            _lib.call("owl_slab_reduce", ops.stream(), bw[slab], grad_w, n_out * n_in, n_out * n_in, ns, accumulate)
            if bs is not None:
                _lib.call("owl_slab_reduce", ops.stream(), bs, grad_b, n_out, n_out, ns, 1)
            return
        tA, tB = (bw["tAh"], bw["tBh"]) if rows == Mh else (bw["tA"], bw["tB"])
        ld = tA.shape[1]
        assert ld == rows_pad
        ops.transpose_colsum(dy, tA, grad_b, rows, n_out, ld_in=dy.shape[-1], ld_out=ld, partials=bw[part])
        ops.transpose_colsum(x, tB, None, rows, n_in, ld_in=x.shape[-1], ld_out=ld)
        want = _split_k(n_out, n_in, rows_pad)
        ns = _lib.load().owl_gemm_effective_splits(rows_pad, want)
        ops.gemm(ops.EPI_SLAB_F32, tA, tB, bw[slab], M=n_out, N=n_in, K=rows_pad, lda=ld, ldw=ld, ldo=n_in,
                 a_rows=n_out, w_rows=n_in, splits=want)
        _lib.call("owl_slab_reduce", ops.stream(), bw[slab], grad_w, n_out * n_in, n_out * n_in, ns, accumulate)

    # The two heads only meet in d(feats): with sub-batch streams on (and every dW on the TN kernel, so that the heads share no transposed-operand
    # scratch) the class head's backward runs on the side stream with its own slab / reduction / transposed-weight scratch, beside the box
    # head's on this one; the box head's last GEMM accumulates into d(feats) behind the class head's event.  Same kernels, same order of the
    # two contributions: same bits.
    # (streams: in-line, the backward shares the forward's side streams; as a deferred tail -- models.OwlViT.overlap_tail -- it has side streams
    #  and fork / join events of its own, because the next forward is using the model's while this runs)
    S, J, fork = model._bwd_streams()
    main0 = torch.cuda.current_stream()
    hs = S(1) if (model.head_streams and model.encoder_streams > 1 and len(model._encoder_chunks(B)) > 1 and bw["tn_all"]) else main0
    ev_h = model._dw_events
    if hs is not main0:
        ev_h[0].record(main0)
        hs.wait_event(ev_h[0])
    cs, cp, cw = ("slab2", "part2", "wT2") if hs is not main0 else ("slab", "part", "wT")
    # ---- class head ---------------------------------------------------------------------------------
    with torch.cuda.stream(hs):
        ops.class_sims_bwd(d_sims, sims, ws["argmax"], ws["inv_norm"], ws["e"], ws["qhat"], bw["de"], bw["g32"], bw["e_bf"], Mh, Dt, C)
        dW(bw["g32"], bw["e_bf"], bw["dqhat"], 32, Dt, Mh, Mhp, None, accumulate=0, part=cp, slab=cs)          # dqhat = G^T e
        _lib.call("owl_query_normalize_bwd", ops.stream(), bw["dqhat"], P_["queries"], G("queries"), cfg.queries, Dt)
        dW(bw["de"], ws["feats"], G("class_predictor.dense0.weight"), Dt, D, Mh, Mhp, G("class_predictor.dense0.bias"), part=cp, slab=cs)
        ops.gemm(ops.EPI_F32, bw["de"], wT("class_predictor.dense0.weight", Dt, D, buf=cw), bw["dfeats"], M=Mh, N=D, K=Dt)
        if hs is not main0:
            ev_h[1].record(hs)
    # ---- box head -------------------------------------------------------------------------------------
    gw2, gb2 = G("box_head.dense2.weight"), G("box_head.dense2.bias")
    assert gb2.data_ptr() == gw2.data_ptr() + 4 * gw2.numel(), "dense2 weight/bias grads must be adjacent in the flat bucket"
    ops.box_final_bwd(d_boxes, ws["sig"], ws["hb1"], ws["ub1"], P_["box_head.dense2.weight"], bw["du1"], bw["box_part"], gw2, Mh, D,
                      du1_colsum=G("box_head.dense1.bias"))          # (dense1's bias gradient from the same pass: no column-sum launch over du1)
    dW(bw["du1"], ws["hb0"], G("box_head.dense1.weight"), D, D, Mh, Mhp, None)
    ops.gemm(ops.EPI_DGELU_BF16, bw["du1"], wT("box_head.dense1.weight", D, D), bw["du0"], aux=ws["ub0"], M=Mh, N=D, K=D)
    dW(bw["du0"], ws["feats"], G("box_head.dense0.weight"), D, D, Mh, Mhp, G("box_head.dense0.bias"))
    w0T = wT("box_head.dense0.weight", D, D)
    if hs is not main0:
        main0.wait_event(ev_h[1])                 # d(feats) of the class head is in place (and the side stream's scratch is free again)
    ops.gemm(ops.EPI_ACC_F32, bw["du0"], w0T, bw["dfeats"], M=Mh, N=D, K=D)
    # ---- merge + the two final LayerNorms --------------------------------------------------------------
    ops.merge_ln_bwd(bw["dfeats"], ws["x_fin"], ws["cls_ln"], ws["st_post"], ws["st_pp"], P_["backbone.post_layernorm.weight"],
                     P_["backbone.post_layernorm.bias"], P_["post_post_layernorm.weight"], bw["dx"], bw["dcls"],
                     G("backbone.post_layernorm.weight"), G("backbone.post_layernorm.bias"), G("post_post_layernorm.weight"),
                     G("post_post_layernorm.bias"), B, P, Tp, D, partials=bw["part"], dx_bf16=bw["dxb"],
                     dx_colsum=G(tl + "mlp.fc2.bias") if cfg.trainable_layer() == cfg.layers - 1 else None)     # (dx here = d(output of the last layer))
    scale = cfg.head_dim ** -0.5
    # (bw["dxb"] always holds the bf16 copy of bw["dx"]: every kernel that writes dx writes it too -- no separate cast pass)
    # ---- frozen layers ABOVE the trainable one (literal "layers.11" rule on a deeper model): dX only ----------
    # Like the encoder forward (models.OwlViT._forward_impl), this chain couples no two images: it runs as sub-batches (row ranges of the
    # same buffers) on the model's streams, layer by layer.
    upper = range(cfg.layers - 1, cfg.trainable_layer(), -1)
    if len(upper) > 0:
        chunks = model._encoder_chunks(B)
        main = torch.cuda.current_stream()
        streams = [main] + [S(c) for c in range(1, len(chunks))]
        if len(chunks) > 1:
            fork.record(main)
            for s_ in streams[1:]:
                s_.wait_event(fork)
        for i in upper:
            Ls, fz = model._layer_ws(B, i), model._fz
            pre = f"backbone.encoder.layers.{i}."
            for (b0, nb), s_ in zip(chunks, streams):
                r0, Mc = b0 * Tp, nb * Tp
                R = lambda t: t[r0:r0 + Mc]
                with torch.cuda.stream(s_):
                    ops.gemm(ops.EPI_DQGELU_BF16, R(bw["dxb"]), fz[f"{i}.w2T"], R(bw["du"]), aux=R(Ls["gp"]), M=Mc, N=I, K=D, concurrency=len(chunks))
                    ops.gemm(ops.EPI_BIAS_BF16, R(bw["du"]), fz[f"{i}.w1T"], R(bw["dh"]), M=Mc, N=D, K=I, concurrency=len(chunks))
                    # (the LayerNorm backward also writes the bf16 copy of its dx: the operand of the next dX GEMM, no separate cast pass)
                    ops.layernorm_bwd(R(bw["dh"]), R(Ls["x_mid"]), R(Ls["st2"]), P_[pre + "layer_norm2.weight"], R(bw["dx"]), R(bw["dxm"]), None, None,
                                      Mc, D, dx_bf16=R(bw["dxb"]))
                    ops.gemm(ops.EPI_BIAS_BF16, R(bw["dxb"]), fz[f"{i}.woT"], R(bw["datt"]), M=Mc, N=D, K=D, concurrency=len(chunks))
                    ops.attention_bwd(R(Ls["qkv"]), R(bw["datt"]), R(Ls["att"]), Ls["lse"][b0:b0 + nb], bw["dvec"][b0:b0 + nb], R(bw["dqkv"]),
                                      nb, H, T, Tp, scale)
                    ops.gemm(ops.EPI_BIAS_BF16, R(bw["dqkv"]), fz[f"{i}.wqkvT"], R(bw["dh"]), M=Mc, N=D, K=3 * D, concurrency=len(chunks))
                    ops.layernorm_bwd(R(bw["dh"]), R(Ls["x_in"]), R(Ls["st1"]), P_[pre + "layer_norm1.weight"], R(bw["dxm"]), R(bw["dx"]), None, None,
                                      Mc, D, dx_bf16=R(bw["dxb"]))
        for c, s_ in enumerate(streams):
            if c > 0:
                J(c).record(s_)
                main.wait_event(J(c))
    Lt = model._layer_ws(B, cfg.trainable_layer())
    # ---- trainable encoder layer ------------------------------------------------------------------------------
    # Two chains: dX (this stream) and the four weight gradients.  A weight gradient feeds nothing downstream -- it only has to be in the
    # bucket when backward() returns -- so the dW GEMMs (+ their bias column sums and slab reductions) run on the model's side stream, each
    # behind the event of the dX-chain kernel that produces its operand: its workgroups fill the CUs the dX kernels' last rounds leave
    # idle, and vice versa.  Same kernels on the same operands: same bits.  The side stream owns the split-K slab from here on and has its
    # own reduction scratch; the second bf16 dx goes to its own buffer because dW(fc2) may still be reading the first.
    main = torch.cuda.current_stream()
    side = S(1) if model.encoder_streams > 1 else main
    evs = model._dw_events

    def on_side(k, fn):
        if side is main:
            fn()
            return
        evs[k].record(main)
        side.wait_event(evs[k])
        with torch.cuda.stream(side):
            fn()

    # MLP
    if cfg.trainable_layer() != cfg.layers - 1:     # (the last layer's fc2 bias gradient came out of merge_ln_bwd)
        ops.colsum_f32(bw["dx"], G(tl + "mlp.fc2.bias"), M, D, partials=bw["part"])
    on_side(0, lambda: dW(bw["dxb"], Lt["g"], G(tl + "mlp.fc2.weight"), D, I, M, Mp, part="part2"))
    dxc = 1 if side is main else 2          # (the weight-gradient GEMMs run beside the dX chain: ops.gemm's small-problem rule counts them)
    ops.gemm(ops.EPI_DQGELU_BF16, bw["dxb"], wT(tl + "mlp.fc2.weight", D, I), bw["du"], aux=Lt["gp"], M=M, N=I, K=D, concurrency=dxc)
    on_side(1, lambda: dW(bw["du"], Lt["h2"], G(tl + "mlp.fc1.weight"), I, D, M, Mp, G(tl + "mlp.fc1.bias"), part="part2"))
    ops.gemm(ops.EPI_BIAS_BF16, bw["du"], wT(tl + "mlp.fc1.weight", I, D), bw["dh"], M=M, N=D, K=I, concurrency=dxc)
    ops.layernorm_bwd(bw["dh"], Lt["x_mid"], Lt["st2"], P_[tl + "layer_norm2.weight"], bw["dx"], bw["dxm"],
                      G(tl + "layer_norm2.weight"), G(tl + "layer_norm2.bias"), M, D, dx_bf16=bw["dxb2"], partials=bw["part"],
                      dx_colsum=G(tl + "self_attn.out_proj.bias"))     # (dx here = d(x + out-proj output): its column sums are that bias's gradient)
    # attention
    on_side(2, lambda: dW(bw["dxb2"], Lt["att"], G(tl + "self_attn.out_proj.weight"), D, D, M, Mp, part="part2"))
    woT = wT(tl + "self_attn.out_proj.weight", D, D)
    ops.gemm(ops.EPI_BIAS_BF16, bw["dxb2"], woT, bw["datt"], M=M, N=D, K=D, concurrency=dxc)
    ops.attention_bwd(Lt["qkv"], bw["datt"], Lt["att"], Lt["lse"], bw["dvec"], bw["dqkv"], B, H, T, Tp,
                      cfg.head_dim ** -0.5)
    o = model.flat_offsets[tl + "self_attn.q_proj.weight"]
    g_wqkv = model.flat_grad[o: o + 3 * D * D].view(3 * D, D)
    ob = model.flat_offsets[tl + "self_attn.q_proj.bias"]
    g_bqkv = model.flat_grad[ob: ob + 3 * D]
    on_side(3, lambda: dW(bw["dqkv"], Lt["h1"], g_wqkv, 3 * D, D, M, Mp, g_bqkv, part="part2"))
    wqkv = model.flat_bf16[o: o + 3 * D * D].view(3 * D, D)
    if pre_wt is not None:
        wqkvT = pre_wt["qkv"]
    else:
        wqkvT = bw["wT"][: 3 * D * D].view(D, 3 * D)
        ops.transpose_bf16(wqkv, wqkvT, 3 * D, D)
    ops.gemm(ops.EPI_BIAS_BF16, bw["dqkv"], wqkvT, bw["dh"], M=M, N=D, K=3 * D, concurrency=dxc)
    # everything below layer_norm1 is frozen: only its affine parameters need gradients
    ops.layernorm_bwd(bw["dh"], Lt["x_in"], Lt["st1"], P_[tl + "layer_norm1.weight"], None, None,
                      G(tl + "layer_norm1.weight"), G(tl + "layer_norm1.bias"), M, D, partials=bw["part"])
    if side is not main:
        evs[4].record(side)
        main.wait_event(evs[4])
