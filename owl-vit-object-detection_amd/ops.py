"""Thin Python wrappers over the C ABI (include/owl_hip.h): shape/dtype validation, stream, call.

PyTorch is plumbing here (device memory + streams); every op below runs a hand-written HIP kernel
from libowlhip.so and raises if the library is missing -- there is no eager fallback.
"""
import torch

from . import _lib

EPI_BIAS_BF16, EPI_QGELU_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32, EPI_ATOMIC_F32 = 0, 1, 2, 3, 4, 5
EPI_TRANS_BF16, EPI_PATCH_F32, EPI_DQGELU_BF16, EPI_DGELU_BF16, EPI_ACC_F32, EPI_SLAB_F32 = 6, 7, 8, 9, 10, 11

ROW_PAD = 128
CHIP_CUS = 256     # MI355X (gfx950), the only target: the kernels' persistent grids and the small-problem rule below count on it (csrc/gemm_common.h NUM_CUS);
                   # models.OwlViT warns when the device reports another count (a partitioned GPU)
ATTN_VARIANT = 0   # default `variant` of attention_fwd_vrow(): 0 = the library's choice; 1 plain tiling, 2 class token peeled (tests / tools)
GEMM_TILE = 0      # default `tile` argument of gemm(): 0 = automatic kernel choice; tests / tools pin one kernel (128 | 256 | 7; tuning builds: 8 | 9 | 5 | 4)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def pad_rows(n: int) -> int:
    return (n + ROW_PAD - 1) // ROW_PAD * ROW_PAD


def zeros_rows(rows: int, width: int, dtype, device) -> torch.Tensor:
    """[pad_rows(rows), width] zero buffer (pad rows stay zero / finite by construction)."""
    return torch.zeros(pad_rows(rows), width, dtype=dtype, device=device)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def gemm(epi, A, W, out, bias=None, resid=None, aux=None, M=None, N=None, K=None, lda=None, ldw=None, ldo=None,
         ld_aux=0, a_rows=None, w_rows=None, alpha=1.0, splits=1, Tp=0, tile=None, concurrency=None):
    """out = epilogue(A[M,K] @ W[N,K]^T).  A/W bf16; see include/owl_hip.h for epilogues.
    concurrency: how many launches like this one the caller has in flight at once (sub-batch streams).  If their 256 x 256 tiles together fill at most half
    the chip's CHIP_CUS compute units, the automatic kernel choice becomes tile 6 (half-height tiles; include/owl_hip.h): the library cannot know what else is running."""
    _chk(A, torch.bfloat16, "A"); _chk(W, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    K = K if K is not None else A.shape[-1]
    N = N if N is not None else W.shape[0]
    M = M if M is not None else A.shape[0]
    lda = lda if lda is not None else A.shape[-1]
    ldw = ldw if ldw is not None else W.shape[-1]
    ldo = ldo if ldo is not None else (out.shape[-1] if epi != EPI_TRANS_BF16 else 0)
    a_rows = a_rows if a_rows is not None else A.shape[0]
    w_rows = w_rows if w_rows is not None else W.shape[0]
    if aux is not None and ld_aux == 0:
        ld_aux = aux.shape[-1]
    tile = int(GEMM_TILE if tile is None else tile)
    if tile == 0 and concurrency is not None and 2 * int(concurrency) * ((M + 255) // 256) * ((N + 255) // 256) <= CHIP_CUS:
        tile = 6
    _lib.call("owl_gemm_nt_bf16", stream(), epi, A, lda, a_rows, W, ldw, w_rows, bias, out, ldo, resid, aux, ld_aux,
              M, N, K, float(alpha), int(splits), int(Tp), tile)
    return out


def patch_embed(image_bf16, w_pe, pos, x_out, B, S, ps, D, Tp, scratch=None, tile=0):
    _chk(image_bf16, torch.bfloat16, "image"); _chk(w_pe, torch.bfloat16, "w_pe"); _chk(pos, torch.float32, "pos")
    _chk(x_out, torch.float32, "x_out")
    _lib.call("owl_patch_embed_bf16", stream(), image_bf16, w_pe, pos, x_out, scratch, B, S, ps, D, Tp, int(tile))


def cls_rows(x, cls, pos, B, Tp, D):
    _lib.call("owl_cls_rows", stream(), x, cls, pos, B, Tp, D)


def layernorm(x, gamma, beta, out, rows, D, stats=None, eps=1e-5, delta=None, x_out=None, delta2=None, store_x=True):
    """out = LN(x) -- or, with `delta` (bf16; optionally a second one), s = x + delta (+ delta2), out = LN(s) and
    x_out = s in one pass; `store_x=False` skips the write of s (the caller re-forms it later from the same operands)."""
    _chk(x, torch.float32, "x"); _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    ob = 1 if out.dtype == torch.bfloat16 else 0
    if delta is None:
        _lib.call("owl_layernorm_fwd", stream(), x, gamma, beta, out, ob, stats, rows, D, float(eps))
    else:
        _chk(delta, torch.bfloat16, "delta"); _chk(delta2, torch.bfloat16, "delta2")
        xo = (x_out if x_out is not None else x) if store_x else None
        _lib.call("owl_add_layernorm_fwd", stream(), x, delta, xo, gamma, beta, out, ob, stats, rows, D, float(eps), delta2)
    return out


def _tuning_only(what):
    if not _lib.is_tuning_build():
        raise RuntimeError(f"{what} exists only in an OWL_TUNING build of libowlhip.so (OWL_TUNING=1 bash csrc/build.sh); the loaded library is the shipped build")


def attention_fwd(q, k, ld_qk, vt, vt_img_stride, out, ld_out, lse, B, H, T, Tp, scale):
    """Tuning builds only: the round-1 form with V^T per head (include/owl_hip_tuning.h)."""
    _tuning_only("the V^T attention forward")
    _lib.call("owl_attention_fwd_bf16", stream(), q, k, ld_qk, vt, vt_img_stride, out, ld_out, lse, B, H, T, Tp,
              float(scale))
    return out


_attn_redo = {}


def attention_redo_ws(B, H, T, device):
    """Tuning builds only: scratch of the one-wave-per-SIMD attention forward (one int per query block; contents irrelevant): one buffer per (device, stream)."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    n = B * H * ((T - 1 + 255) // 256 + 1)
    buf = _attn_redo.get(key)
    if buf is None or buf.numel() < n:
        buf = _attn_redo[key] = torch.zeros(max(n, 4096) + 8192, dtype=torch.int32, device=device)
    return buf


ATTN_SLOW_TILES = None   # optional int32[1] device tensor: every attention forward adds its count of slow-path tiles (bench.py --weights trained_like, tests)


def attention_fwd_vrow(q, k, v, ld_qkv, out, ld_out, lse, B, H, T, Tp, scale, variant=None, slow_tiles=None):
    """Fused attention forward; q / k / v are column slices of the row-major qkv rows (no V^T copy).  variant 0-2: include/owl_hip.h; 3-5: tuning builds."""
    variant = int(ATTN_VARIANT if variant is None else variant)
    if variant >= 3:
        _tuning_only(f"attention forward variant {variant}")
        _lib.call("owl_attention_fwd_w64_bf16", stream(), q, k, v, ld_qkv, out, ld_out, lse, B, H, T, Tp, float(scale), variant, attention_redo_ws(B, H, T, out.device))
        return out
    if slow_tiles is None:
        slow_tiles = ATTN_SLOW_TILES
    _lib.call("owl_attention_fwd_vrow_bf16", stream(), q, k, v, ld_qkv, out, ld_out, lse, B, H, T, Tp, float(scale), variant, slow_tiles)
    return out


def merge_ln(x, g1, b1, g2, b2, cls_ln, feats, stats1, stats2, B, P, Tp, D, eps=1e-5, delta=None, x_out=None):
    _lib.call("owl_merge_ln_fwd", stream(), x, delta, (x_out if x_out is not None else x) if delta is not None else None,
              g1, b1, g2, b2, cls_ln, feats, stats1, stats2, B, P, Tp, D, float(eps))


def query_normalize(queries, qhat32, qnorm, nq, Dt):
    _chk(queries, torch.float32, "queries")
    _lib.call("owl_query_normalize", stream(), queries, qhat32, qnorm, nq, Dt)


def class_sims(e, qhat32, sims, argmax, inv_norm, rows, Dt, C):
    _chk(e, torch.float32, "e")
    _lib.call("owl_class_sims_fwd", stream(), e, qhat32, sims, argmax, inv_norm, rows, Dt, C)


def box_final(h, w2, b2, box_bias, boxes, sig, rows, P, D):
    _chk(h, torch.bfloat16, "h"); _chk(w2, torch.float32, "w2")
    _lib.call("owl_box_final_fwd", stream(), h, w2, b2, box_bias, boxes, sig, rows, P, D)


def cast_bf16(src, dst=None):
    _chk(src, torch.float32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    _lib.call("owl_cast_f32_bf16", stream(), src, dst, src.numel())
    return dst


def transpose_bf16(src, dst, R, C, ld_in=None, ld_out=None):
    _lib.call("owl_transpose_bf16", stream(), src, ld_in if ld_in is not None else src.shape[-1], dst,
              ld_out if ld_out is not None else dst.shape[-1], R, C)
    return dst


# ---- backward-side wrappers ---------------------------------------------------------------------------
_partials = {}


def rowreduce_workspace(groups, rows_per_group, C, device):
    """f32 partial-sum scratch for the deterministic row reductions (owl_rowreduce_workspace_bytes)."""
    nbytes = torch.zeros(1, dtype=torch.int64)
    _lib.call("owl_rowreduce_workspace_bytes", int(groups), int(rows_per_group), int(C), nbytes)
    return torch.empty(int(nbytes.item()) // 4, dtype=torch.float32, device=device)


def _part(partials, rows, C, device):
    """The caller's scratch, or (stand-alone use: tests, tools) a cached per-device buffer grown on demand."""
    if partials is not None:
        return partials
    need = ((int(rows) + 63) // 64 + 1) * 6 * max(int(C), 4)
    buf = _partials.get(device)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.float32, device=device)
        _partials[device] = buf
    return buf


def layernorm_bwd(dy, x, stats, gamma, dres, dx, dgamma, dbeta, rows, D, dx_bf16=None, partials=None, dx_colsum=None):
    """dx (f32) = LN backward (+ dres); optionally also its bf16 copy `dx_bf16` (the operand of the next dX GEMM) and `dx_colsum` +=
    the column sums of dx (the bias gradient of the linear layer in front of this residual position)."""
    _chk(dx_bf16, torch.bfloat16, "dx_bf16"); _chk(dx_colsum, torch.float32, "dx_colsum")
    part = _part(partials, rows, D, x.device) if dgamma is not None else None
    _lib.call("owl_layernorm_bwd", stream(), dy, 1 if dy.dtype == torch.bfloat16 else 0, x, stats, gamma, dres, dx,
              dgamma, dbeta, rows, D, dx_bf16, part, part.numel() if part is not None else 0, dx_colsum)


def merge_ln_bwd(dfeats, x, cls_ln, st1, st2, g1, b1, g2, dx, dcls_ws, dg1, db1, dg2, db2, B, P, Tp, D, partials=None, dx_bf16=None, dx_colsum=None):
    """dx_colsum (optional): += column sums of dx over all tokens (the bias gradient of the last layer's fc2) in the same pass."""
    part = partials if partials is not None else _part(None, B * ((P + 63) // 64) * 64, D, x.device)
    _chk(dx_bf16, torch.bfloat16, "dx_bf16"); _chk(dx_colsum, torch.float32, "dx_colsum")
    _lib.call("owl_merge_ln_bwd", stream(), dfeats, x, cls_ln, st1, st2, g1, b1, g2, dx, dcls_ws, dg1, db1, dg2, db2, B, P, Tp, D,
              part, part.numel(), dx_bf16, dx_colsum)


def class_sims_bwd(dsims, sims, argmax, inv_norm, e, qhat32, de, g32, e_bf16, rows, Dt, C):
    _lib.call("owl_class_sims_bwd", stream(), dsims, sims, argmax, inv_norm, e, qhat32, de, g32, e_bf16, rows, Dt, C)


def box_final_bwd(dboxes, sig, h1, u1, w2, du1, partials, dw2_db2, rows, D, du1_colsum=None):
    """partials: f32 [owl_box_final_bwd_blocks(rows), 5 * D + 4]; du1_colsum (optional): += column sums of du1 (dense1's bias gradient)."""
    if partials.numel() < _lib.load().owl_box_final_bwd_blocks(int(rows)) * (5 * int(D) + 4):
        raise ValueError("box_final_bwd: partials must hold owl_box_final_bwd_blocks(rows) x (5 D + 4) floats")
    _chk(du1_colsum, torch.float32, "du1_colsum")
    _lib.call("owl_box_final_bwd", stream(), dboxes, sig, h1, u1, w2, du1, partials, dw2_db2, rows, D, du1_colsum)


def transpose_colsum(src, dst, colsum, R, C, ld_in=None, ld_out=None, partials=None):
    part = _part(partials, R, C, src.device) if colsum is not None else None
    _lib.call("owl_transpose_colsum_bf16", stream(), src, ld_in if ld_in is not None else src.shape[-1], dst,
              (ld_out if ld_out is not None else dst.shape[-1]) if dst is not None else 0, colsum, R, C,
              part, part.numel() if part is not None else 0)


def colsum_f32(src, colsum, R, C, partials=None):
    part = _part(partials, R, C, src.device)
    _lib.call("owl_colsum_f32", stream(), src, colsum, R, C, part, part.numel())


def attention_bwd(qkv, dO, O, lse, dvec, dqkv, B, H, T, Tp, scale, phases=0):
    """phases: 0 = dvec + dK / dV + dQ on this stream; a mask (1 | 2 | 4) launches a subset (dK / dV and dQ only share inputs: tools/attn_bwd_overlap_ab.py)."""
    _lib.call("owl_attention_bwd_bf16", stream(), qkv, dO, O, lse, dvec, dqkv, B, H, T, Tp, float(scale), int(phases))


_pp_ws = {}


NMS_ROUTES = {"per_class": 0, "coordinate_offset": 1, "torchvision_gpu": 2, "torchvision_cpu": 3}


def postprocess(boxes, sims, max_out, conf_thr, iou_thr, route="torchvision_gpu"):
    """owl_postprocess: boxes [B,P,4] f32, sims [B,P,C] f32 -> (boxes [B,K,4], classes [B,K] i64 (-1 pad), scores [B,K]
    (0 pad), patch [B,K] i64 (-1 pad), counts [B] i32) with K = max_out.  ref src/models.py:127-146."""
    B, P, C = sims.shape
    dev = sims.device
    key = (B, P, dev)
    if key not in _pp_ws:
        nbytes = torch.zeros(1, dtype=torch.int64)
        _lib.call("owl_postprocess_workspace", B, P, nbytes)
        _pp_ws[key] = torch.empty(int(nbytes.item()), dtype=torch.uint8, device=dev)
    ws = _pp_ws[key]
    out_boxes = torch.zeros(B, max_out, 4, dtype=torch.float32, device=dev)
    out_scores = torch.zeros(B, max_out, dtype=torch.float32, device=dev)
    out_classes = torch.full((B, max_out), -1, dtype=torch.int64, device=dev)
    out_patch = torch.full((B, max_out), -1, dtype=torch.int64, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    _lib.call("owl_postprocess", stream(), boxes, sims, ws, ws.numel(), out_boxes, out_scores, out_classes, out_patch, counts,
              B, P, C, max_out, conf_thr, iou_thr, NMS_ROUTES[route])
    return out_boxes, out_classes, out_scores, out_patch, counts


_zero_row = {}


def gemm_tn_slab(dy, x, slab, rows, n_out, n_in, splits, variant=0, bias_slab=None):
    """slab[s][n][k] = sum_{m in split s} dy[m][n] * x[m][k] (weight gradient, no transposed copies) -> splits used.
    bias_slab (optional, f32 [>= splits, n_out]): also bias_slab[s][n] = sum_{m in split s} dy[m][n] from the same pass (the bias gradient's partial sums)."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(slab, torch.float32, "slab"); _chk(bias_slab, torch.float32, "bias_slab")
    dev = dy.device
    if dev not in _zero_row:
        _zero_row[dev] = torch.zeros(512, dtype=torch.bfloat16, device=dev)
    used = torch.zeros(1, dtype=torch.int32)
    _lib.call("owl_gemm_tn_slab_bf16", stream(), dy, dy.shape[-1], x, x.shape[-1], _zero_row[dev], slab, rows, n_out, n_in,
              int(splits), used, int(variant), bias_slab)
    return int(used.item())


def colsum_bf16(src, colsum, rows, cols, partials=None):
    _chk(src, torch.bfloat16, "src"); _chk(colsum, torch.float32, "colsum")
    part = _part(partials, rows, cols, src.device)
    _lib.call("owl_colsum_bf16", stream(), src, src.shape[-1], colsum, rows, cols, part, part.numel())
