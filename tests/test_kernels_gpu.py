"""Per-kernel numerics on a real MI355X: each HIP kernel (called through the C ABI) against a plain
PyTorch fp32 reference of the same op on the same (bf16-rounded) inputs."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from owl_vit_object_detection_amd import ops  # noqa: E402


def _TUNING_BUILD():
    """Is the loaded libowlhip.so an OWL_TUNING build?  Asked of the library, not of the environment."""
    from owl_vit_object_detection_amd import _lib as _L
    try:
        return _L.is_tuning_build()
    except _L.OwlLibError:
        return False


DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def report(name, got, ref, atol, rtol):
    got = got.float(); ref = ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} off; max err {float(err.max()):.4g} "
                             f"(ref max {float(ref.abs().max()):.4g}); first bad idx {idx}; "
                             f"got {got[bad][:4].tolist()} ref {ref[bad][:4].tolist()}")


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


_TUNING = _TUNING_BUILD()


@pytest.fixture(params=[128, 256, 7, 0] + ([8, 5] if _TUNING else []), ids=["tile128", "tile256", "pingpong2", "auto"] + (["pingpong", "free-running"] if _TUNING else []))
def gemm_tile(request):
    """Run the GEMM tests on every kernel: 128x128 4-wave, 256x256 8-wave (the single-phase reference kernels), the two-phase ping-pong kernel on
    the whole problem (gemm_pp2.hip: what the model runs) and the library's automatic choice (two-phase + half-height remainder tiles,
    gemm_pph.hip) -- each against `A @ W.T` in fp32, including the odd shapes (M = 777, N = 264 / 328) where clamped loads and store guards
    live.  Tuning builds add the round-1 four-phase ping-pong kernel (gemm_pp.hip) and the round-4 free-running one (gemm_fr.hip)."""
    from owl_vit_object_detection_amd import _lib
    ops.GEMM_TILE = request.param
    yield request.param
    ops.GEMM_TILE = 0


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (2048, 768, 768), (128, 128, 64), (1000, 3072, 256), (1000, 328, 256), (777, 264, 128)])
def test_gemm_bias_bf16(M, N, K, gemm_tile):
    A = ops.zeros_rows(M, K, torch.bfloat16, DEV)
    A[:M] = rnd(M, K).bfloat16()
    W = rnd(N, K, scale=0.1, seed=1).bfloat16()
    bias = rnd(N, seed=2)
    out = ops.zeros_rows(M, N, torch.bfloat16, DEV)
    ops.gemm(ops.EPI_BIAS_BF16, A, W, out, bias=bias, M=M)
    ref = A[:M].float() @ W.float().t() + bias
    report("gemm bias", out[:M], ref, 2e-2, 1e-2)
    assert float(out[M:].abs().max()) == 0.0 if out.shape[0] > M else True


def test_gemm_epilogues(gemm_tile):
    M, N, K = 300, 256, 192
    A = ops.zeros_rows(M, K, torch.bfloat16, DEV); A[:M] = rnd(M, K).bfloat16()
    W = rnd(N, K, scale=0.1, seed=1).bfloat16()
    bias = rnd(N, seed=2)
    acc = A[:M].float() @ W.float().t()
    u = acc + bias
    # quick-gelu; saved for the backward: its derivative at the f32 pre-activation (round 6; before: the pre-activation itself)
    out = ops.zeros_rows(M, N, torch.bfloat16, DEV); aux = ops.zeros_rows(M, N, torch.bfloat16, DEV)
    ops.gemm(ops.EPI_QGELU_BF16, A, W, out, bias=bias, aux=aux, M=M)
    sg = torch.sigmoid(1.702 * u)
    report("qgelu", out[:M], qgelu(u), 2e-2, 1e-2); report("qgelu aux = quick_gelu'(u)", aux[:M], sg * (1.0 + 1.702 * u * (1.0 - sg)), 5e-3, 5e-3)
    out2 = ops.zeros_rows(M, N, torch.bfloat16, DEV)
    ops.gemm(ops.EPI_QGELU_BF16, A, W, out2, bias=bias, M=M)
    assert torch.equal(out, out2)                       # saving the derivative does not touch the output's bits
    # erf gelu
    out.zero_()
    ops.gemm(ops.EPI_GELU_BF16, A, W, out, bias=bias, M=M)
    report("gelu", out[:M], F.gelu(u), 2e-2, 1e-2)
    # residual f32 (in place)
    x = ops.zeros_rows(M, N, torch.float32, DEV); x[:M] = rnd(M, N, seed=5)
    ref = x[:M].clone() + u
    ops.gemm(ops.EPI_RESID_F32, A, W, x, bias=bias, resid=x, M=M)
    report("resid", x[:M], ref, 1e-3, 1e-4)
    # f32 with alpha, no bias
    o32 = ops.zeros_rows(M, N, torch.float32, DEV)
    ops.gemm(ops.EPI_F32, A, W, o32, M=M, alpha=0.5)
    report("f32 alpha", o32[:M], 0.5 * acc, 1e-3, 1e-4)
    # accumulate
    ops.gemm(ops.EPI_ACC_F32, A, W, o32, M=M)
    report("acc", o32[:M], 1.5 * acc, 1e-3, 1e-4)
    # split-K atomic (tuning builds only: the train path uses the deterministic slabs below)
    if _TUNING_BUILD():
        o32.zero_()
        ops.gemm(ops.EPI_ATOMIC_F32, A, W, o32, M=M, splits=3)
        report("atomic", o32[:M], acc, 1e-3, 1e-4)
    # split-K slabs + deterministic reduce
    from owl_vit_object_detection_amd import _lib
    ns = _lib.load().owl_gemm_effective_splits(K, 2)
    slabs = torch.zeros(ns, M, N, device=DEV)
    ops.gemm(ops.EPI_SLAB_F32, A, W, slabs, M=M, splits=2, ldo=N)
    tgt = torch.ones(M, N, device=DEV)
    _lib.call("owl_slab_reduce", ops.stream(), slabs, tgt, M * N, M * N, ns, 1)
    report("slab", tgt, acc + 1.0, 1e-3, 1e-4)
    # activation-derivative epilogues
    upre = ops.zeros_rows(M, N, torch.bfloat16, DEV); upre[:M] = rnd(M, N, seed=9).bfloat16()
    uf = upre[:M].float()
    uf.requires_grad_(True)
    (g2,) = torch.autograd.grad(F.gelu(uf).sum(), uf)
    out.zero_()
    ops.gemm(ops.EPI_DQGELU_BF16, A, W, out, aux=upre, M=M)       # out = acc * aux: aux IS the derivative the forward epilogue saved
    report("dqgelu (multiply by the saved derivative)", out[:M], acc * upre[:M].float(), 2e-2, 1e-2)
    # ... and the pair end to end: forward saves quick_gelu'(u), backward multiplies -> d(u) of the autograd reference
    uu = (A[:M].float() @ W.float().t() + bias).detach().requires_grad_(True)
    (g1,) = torch.autograd.grad(qgelu(uu).sum(), uu)
    ops.gemm(ops.EPI_DQGELU_BF16, A, W, out, aux=aux, M=M)
    report("dqgelu, forward-saved derivative", out[:M], acc * g1, 2e-2, 1e-2)
    ops.gemm(ops.EPI_DGELU_BF16, A, W, out, aux=upre, M=M)
    report("dgelu", out[:M], acc * g2, 2e-2, 1e-2)


def test_patch_embed(gemm_tile):
    B, S, ps, D = 3, 96, 16, 128
    G = S // ps; P = G * G; T = P + 1; Tp = (T + 7) // 8 * 8
    img = rnd(B, 3, S, S).bfloat16()
    w = rnd(D, 3, ps, ps, scale=0.05, seed=1).bfloat16()
    pos = rnd(T, D, seed=2); cls = rnd(D, seed=3)
    x = ops.zeros_rows(B * Tp, D, torch.float32, DEV)
    ops.patch_embed(img, w.view(D, -1).contiguous(), pos, x, B, S, ps, D, Tp)
    ops.cls_rows(x, cls, pos, B, Tp, D)
    pe = F.conv2d(img.float(), w.float(), stride=ps).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.expand(B, 1, D), pe], 1) + pos
    got = x[: B * Tp].view(B, Tp, D)[:, :T]
    report("patch embed", got, ref, 1e-3, 1e-3)
    assert float(x[: B * Tp].view(B, Tp, D)[:, T:].abs().max()) == 0.0


@pytest.mark.parametrize("ps,S,D,B,tile", [(14, 84, 128, 3, 0), (14, 336, 256, 2, 7), (14, 336, 256, 2, 0), (24, 96, 128, 2, 7), (16, 96, 256, 6, 7)])
def test_patch_embed_any_patch_size_is_im2row_free(ps, S, D, B, tile):
    """Patch sizes that are not 2^n (L/14: 14-pixel rows = 28 bytes): the ping-pong kernel's A loader gathers 16-byte chunks of the patch rows straight from the
    image -- the K index pads a row to 16 positions, the last chunk overlaps its predecessor instead of leaving the row, the weight holds zeros at the
    duplicated positions (weights.patch_weight_gather_layout) -- against the convolution itself (HF5:282-288); the single-phase kernels (tile 0 on a small
    problem) take an explicit im2row in the same K order."""
    from owl_vit_object_detection_amd import weights
    G = S // ps; P = G * G; T = P + 1; Tp = (T + 7) // 8 * 8
    img = rnd(B, 3, S, S).bfloat16()
    w = rnd(D, 3, ps, ps, scale=0.05, seed=1).bfloat16()
    pos = rnd(T, D, seed=2)
    wk = weights.patch_weight_gather_layout(w, ps).contiguous()
    psp = 16 if ps == 14 else (32 if ps == 24 else ps)
    assert wk.shape == (D, (3 * ps * psp + 63) // 64 * 64)
    x = ops.zeros_rows(B * Tp, D, torch.float32, DEV)
    scratch = ops.zeros_rows(B * P, wk.shape[1], torch.bfloat16, DEV) if (tile == 0 and ps & (ps - 1)) else None
    ops.patch_embed(img, wk, pos, x, B, S, ps, D, Tp, scratch=scratch, tile=tile)
    pe = F.conv2d(img.float(), w.float(), stride=ps).flatten(2).transpose(1, 2) + pos[1:]
    got = x[: B * Tp].view(B, Tp, D)[:, 1:T]
    report(f"patch embed ps={ps} tile={tile}", got, pe, 1e-3, 1e-3)
    if tile == 0 and scratch is None and ps & (ps - 1):
        pytest.fail("unreachable")


@pytest.mark.parametrize("D", [128, 768, 1024])
def test_layernorm(D):
    rows = 77
    x = rnd(rows, D, scale=2.0) + 0.3
    g = 1 + 0.1 * rnd(D, seed=1); b = 0.1 * rnd(D, seed=2)
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    out = torch.zeros(rows, D, dtype=torch.bfloat16, device=DEV); stats = torch.zeros(rows, 2, device=DEV)
    ops.layernorm(x, g, b, out, rows, D, stats)
    report("ln bf16", out, ref, 2e-2, 1e-2)
    report("ln mean", stats[:, 0], x.mean(-1), 1e-5, 1e-5)
    report("ln rstd", stats[:, 1], 1 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5), 1e-5, 1e-4)
    x2 = x.clone()
    ops.layernorm(x2, g, b, x2, rows, D)          # in place f32
    report("ln f32 inplace", x2, ref, 1e-5, 1e-5)
    # fused residual add: x_out = x + delta, out = LN(x_out)
    delta = (0.5 * rnd(rows, D, seed=7)).bfloat16()
    x3 = x.clone(); out3 = torch.zeros(rows, D, dtype=torch.bfloat16, device=DEV)
    ops.layernorm(x3, g, b, out3, rows, D, delta=delta)            # x3 updated in place
    xs = x + delta.float()
    report("add-ln x_out", x3, xs, 1e-6, 1e-6)
    report("add-ln out", out3, F.layer_norm(xs, (D,), g, b, 1e-5), 2e-2, 1e-2)


def test_merge_ln():
    B, P, D = 2, 36, 128
    T = P + 1; Tp = 40
    x = torch.zeros(B * Tp, D, device=DEV); xv = x.view(B, Tp, D)
    xv[:, :T] = rnd(B, T, D, scale=1.5)
    g1 = 1 + 0.1 * rnd(D, seed=1); b1 = 0.1 * rnd(D, seed=2); g2 = 1 + 0.1 * rnd(D, seed=3); b2 = 0.1 * rnd(D, seed=4)
    y = F.layer_norm(xv[:, :T], (D,), g1, b1, 1e-5)
    ref = F.layer_norm(y[:, 1:] * y[:, :1], (D,), g2, b2, 1e-5)
    cls_ln = torch.zeros(B, D, device=DEV); feats = ops.zeros_rows(B * P, D, torch.bfloat16, DEV)
    s1 = torch.zeros(B * Tp, 2, device=DEV); s2 = torch.zeros(B * P, 2, device=DEV)
    ops.merge_ln(x, g1, b1, g2, b2, cls_ln, feats, s1, s2, B, P, Tp, D)
    report("merge feats", feats[: B * P].view(B, P, D), ref, 2e-2, 1e-2)
    report("cls_ln", cls_ln, y[:, 0], 1e-5, 1e-5)
    # with the fused final residual add
    delta = torch.zeros(B * Tp, D, dtype=torch.bfloat16, device=DEV)
    delta.view(B, Tp, D)[:, :T] = (0.3 * rnd(B, T, D, seed=9)).bfloat16()
    xo = torch.zeros_like(x)
    ops.merge_ln(x, g1, b1, g2, b2, cls_ln, feats, s1, s2, B, P, Tp, D, delta=delta, x_out=xo)
    xs = xv[:, :T] + delta.view(B, Tp, D)[:, :T].float()
    y2 = F.layer_norm(xs, (D,), g1, b1, 1e-5)
    report("merge+add x_out", xo.view(B, Tp, D)[:, :T], xs, 1e-6, 1e-6)
    report("merge+add feats", feats[: B * P].view(B, P, D), F.layer_norm(y2[:, 1:] * y2[:, :1], (D,), g2, b2, 1e-5), 2e-2, 1e-2)


@pytest.mark.parametrize("B,P,D", [(2, 36, 128), (3, 100, 768), (1, 67, 1024), (2, 2304, 768)])
def test_merge_ln_bwd_matches_torch_autograd(B, P, D):
    """Backward of `LN2(LN1(x)[1:] * LN1(x)[0])` (ref models.py:80-86): dx for every token (patch rows and the class-token row, f32 + the bf16 copy)
    and the four LayerNorm parameter gradients accumulated into the bucket; ragged row counts (P % 64, P % 4 != 0), both model widths."""
    T = P + 1; Tp = (T + 7) // 8 * 8
    x = torch.zeros(B * Tp, D, device=DEV); xv = x.view(B, Tp, D)
    xv[:, :T] = rnd(B, T, D, scale=1.5)
    g1 = 1 + 0.1 * rnd(D, seed=1); b1 = 0.1 * rnd(D, seed=2); g2 = 1 + 0.1 * rnd(D, seed=3); b2 = 0.1 * rnd(D, seed=4)
    cls_ln = torch.zeros(B, D, device=DEV); feats = ops.zeros_rows(B * P, D, torch.bfloat16, DEV)
    s1 = torch.zeros(B * Tp, 2, device=DEV); s2 = torch.zeros(B * P, 2, device=DEV)
    ops.merge_ln(x, g1, b1, g2, b2, cls_ln, feats, s1, s2, B, P, Tp, D)
    dfeats = rnd(B * P, D, seed=7)
    xr = xv[:, :T].clone().requires_grad_(True)
    G1, B1, G2, B2 = (t.clone().requires_grad_(True) for t in (g1, b1, g2, b2))
    y = F.layer_norm(xr, (D,), G1, B1, 1e-5)
    F.layer_norm(y[:, 1:] * y[:, :1], (D,), G2, B2, 1e-5).backward(dfeats.view(B, P, D))
    dx = torch.zeros(B * Tp, D, device=DEV); dxb = torch.zeros(B * Tp, D, device=DEV, dtype=torch.bfloat16); dcls = torch.zeros(B, D, device=DEV)
    grads = [torch.ones(D, device=DEV) for _ in range(4)]              # accumulated onto (the bucket is not zero in general)
    colsum = torch.ones(D, device=DEV)                                 # += column sums of dx over every token (the last fc2's bias gradient)
    ops.merge_ln_bwd(dfeats, x, cls_ln, s1, s2, g1, b1, g2, dx, dcls, *grads, B, P, Tp, D, dx_bf16=dxb, dx_colsum=colsum)
    scale = float(xr.grad.abs().max())
    ref_cs = xr.grad.sum((0, 1))
    report("dx column sums", colsum - 1.0, ref_cs, 1e-4 * float(ref_cs.abs().max()) + 2e-5 * scale * (B * T) ** 0.5, 1e-4)
    report("dx", dx.view(B, Tp, D)[:, :T], xr.grad, 2e-5 * scale, 1e-4)
    assert torch.equal(dxb.view(B, Tp, D)[:, :T], dx.view(B, Tp, D)[:, :T].bfloat16())
    for name, got, ref in zip(("dg1", "db1", "dg2", "db2"), grads, (G1.grad, B1.grad, G2.grad, B2.grad)):
        report(name, got - 1.0, ref, 1e-4 * float(ref.abs().max()) + 1e-5, 1e-4)
    dx2 = torch.zeros_like(dx); grads2 = [torch.ones(D, device=DEV) for _ in range(4)]
    colsum2 = torch.ones(D, device=DEV)
    ops.merge_ln_bwd(dfeats, x, cls_ln, s1, s2, g1, b1, g2, dx2, dcls, *grads2, B, P, Tp, D, dx_bf16=dxb, dx_colsum=colsum2)
    assert torch.equal(dx, dx2) and torch.equal(colsum, colsum2) and all(torch.equal(a, b) for a, b in zip(grads, grads2))      # fixed-order sums


def _attn_case(B, H, T, seed):
    Tp = (T + 7) // 8 * 8
    D = H * 64
    M = B * Tp
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    v = qkv[:M].view(B, Tp, 3 * D)
    v[:, :T] = rnd(B, T, 3 * D, scale=1.0, seed=seed).bfloat16()
    q = v[:, :T, :D].float().view(B, T, H, 64).transpose(1, 2)
    k = v[:, :T, D:2 * D].float().view(B, T, H, 64).transpose(1, 2)
    vv = v[:, :T, 2 * D:].float().view(B, T, H, 64).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(2, 3) * 0.125, -1)
    ref = (att @ vv).transpose(1, 2).reshape(B, T, D)
    lse_ref = torch.logsumexp(q @ k.transpose(2, 3) * 0.125, -1) / math.log(2.0)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV)
    lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=1)      # plain tiling: tokens 0..T-1 in 64-key tiles
    report(f"attn T={T}", out[:M].view(B, Tp, D)[:, :T], ref, 2e-2, 2e-2)
    report(f"lse T={T}", lse[:, :, :T], lse_ref, 2e-3, 1e-3)


@pytest.mark.parametrize("B,H,T", [(2, 2, 37), (1, 3, 200), (2, 1, 577), (1, 12, 2305)])
def test_attention_fwd(B, H, T):
    _attn_case(B, H, T, seed=T)


def test_attention_fwd_spiked_scores():
    """Large score spread: one key dominating forces big running-max jumps between KV tiles."""
    B, H, T = 1, 1, 300
    Tp, D, M = 304, 64, 304
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    v = qkv[:M].view(B, Tp, 3 * D)
    x = rnd(B, T, 3 * D, seed=3)
    x[0, 250, D:2 * D] *= 12.0      # key 250 spikes (third KV tile)
    x[0, 10, :D] *= 6.0
    v[:, :T] = x.bfloat16()
    q = v[:, :T, :D].float(); k = v[:, :T, D:2 * D].float(); vv = v[:, :T, 2 * D:].float()
    ref = torch.softmax(q @ k.transpose(1, 2) * 0.125, -1) @ vv
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, variant=1)
    report("attn spiked", out[:T], ref[0], 3e-2, 2e-2)


@pytest.mark.parametrize("Dt,C", [(64, 4), (512, 10), (768, 10), (128, 1)])
def test_class_sims(Dt, C):
    rows = 300
    e = rnd(rows, Dt, scale=0.7)
    Q = rnd(3 * C, Dt, seed=4)
    qhat = torch.zeros(32, Dt, device=DEV); qn = torch.zeros(32, device=DEV)
    ops.query_normalize(Q, qhat, qn, 3 * C, Dt)
    qh_ref = Q / torch.linalg.norm(Q, dim=-1, keepdim=True) + 1e-6
    report("qhat", qhat[: 3 * C], qh_ref, 1e-6, 1e-5)
    assert float(qhat[3 * C:].abs().max()) == 0.0
    en = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
    s = en @ qh_ref.t()
    ref = F.max_pool1d(s[None], 3, 3)[0]
    sims = torch.zeros(rows, C, device=DEV); am = torch.zeros(rows, C, dtype=torch.uint8, device=DEV)
    inv = torch.zeros(rows, device=DEV)
    ops.class_sims(e, qhat, sims, am, inv, rows, Dt, C)
    report("sims", sims, ref, 2e-5, 1e-4)
    report("argmax", am.float(), s.view(rows, C, 3).argmax(-1).float(), 0, 0)
    report("inv_norm", inv, 1 / (torch.linalg.norm(e, dim=-1) + 1e-6), 1e-6, 1e-5)


@pytest.mark.parametrize("rows,Dt,C", [(300, 64, 4), (300, 512, 10), (2304, 512, 10), (1000, 768, 10), (73728, 512, 10), (37, 768, 3)])
def test_class_sims_forward_backward_match_torch_autograd(rows, Dt, C):
    """Class head tail (ref src/models.py:24-38) forward AND backward against torch autograd of the same f32 expression: sims / argmax / saved 1/norm, then
    de (bf16), the routed upstream G (bf16 [rows, 32]) and the bf16 copy of e.  Ragged row counts (partial last wave / workgroup), the generic backward
    kernel (Dt = 64) and the software-pipelined one (the model widths 512 / 768), one workgroup per CU at the headline row count."""
    e = rnd(rows, Dt, scale=0.7, seed=1)
    Q = rnd(3 * C, Dt, seed=4)
    qhat = torch.zeros(32, Dt, device=DEV); qn = torch.zeros(32, device=DEV)
    ops.query_normalize(Q, qhat, qn, 3 * C, Dt)
    sims = torch.zeros(rows, C, device=DEV); am = torch.zeros(rows, C, dtype=torch.uint8, device=DEV); inv = torch.zeros(rows, device=DEV)
    ops.class_sims(e, qhat, sims, am, inv, rows, Dt, C)
    er = e.clone().requires_grad_(True)
    qh = qhat[: 3 * C].clone().requires_grad_(True)
    s_all = (er / (torch.linalg.norm(er, dim=-1, keepdim=True) + 1e-6)) @ qh.t()
    ref = F.max_pool1d(s_all[None], 3, 3)[0]
    report("sims", sims, ref.detach(), 2e-5, 1e-4)
    assert torch.equal(am.long(), s_all.detach().view(rows, C, 3).argmax(-1))
    dsims = rnd(rows, C, seed=7)
    ref.backward(dsims)
    de = torch.full((rows, Dt), 7.0, device=DEV, dtype=torch.bfloat16); G = torch.full((rows, 32), 7.0, device=DEV, dtype=torch.bfloat16)
    eb = torch.full((rows, Dt), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.class_sims_bwd(dsims, sims, am, inv, e, qhat, de, G, eb, rows, Dt, C)
    assert torch.equal(eb, e.bfloat16())
    report("de", de, er.grad, 2e-3 * float(er.grad.abs().max()), 8e-3)
    # G[r, j] = dsims[r, c] * inv[r] where j is class c's arg-max prompt, 0 elsewhere; dqhat = G^T e reproduces autograd's prompt gradient
    Gref = torch.zeros(rows, 32, device=DEV)
    Gref.scatter_(1, (3 * torch.arange(C, device=DEV)[None] + am.long()), dsims * inv[:, None])
    report("G", G, Gref, 1e-6, 8e-3)
    report("dqhat", G.float().t()[: 3 * C] @ eb.float(), qh.grad, 2e-2 * float(qh.grad.abs().max()), 2e-2)
    de2 = torch.zeros_like(de); G2 = torch.zeros_like(G); eb2 = torch.zeros_like(eb)
    ops.class_sims_bwd(dsims, sims, am, inv, e, qhat, de2, G2, eb2, rows, Dt, C)
    assert torch.equal(de, de2) and torch.equal(G, G2) and torch.equal(eb, eb2)


def test_box_final():
    B, P, D = 2, 36, 128
    rows = B * P
    h = rnd(rows, D).bfloat16()
    w2 = rnd(4, D, scale=0.1, seed=1); b2 = rnd(4, seed=2); bb = rnd(P, 4, seed=3)
    boxes = torch.zeros(rows, 4, device=DEV); sig = torch.zeros(rows, 4, device=DEV)
    ops.box_final(h, w2, b2, bb, boxes, sig, rows, P, D)
    s = torch.sigmoid(h.float() @ w2.t() + b2 + bb.repeat(B, 1))
    ref = torch.stack([s[:, 0] - 0.5 * s[:, 2], s[:, 1] - 0.5 * s[:, 3], s[:, 0] + 0.5 * s[:, 2], s[:, 1] + 0.5 * s[:, 3]], -1)
    report("sig", sig, s, 1e-5, 1e-5); report("boxes", boxes, ref, 1e-5, 1e-5)


@pytest.mark.parametrize("rows,D", [(72, 128), (2304, 768), (1000, 1024), (73728, 768)])
def test_box_final_bwd_matches_torch_autograd(rows, D):
    """d(xyxy) -> d(cx,cy,w,h) -> sigmoid' -> dense2 backward fused with dense1's erf-GELU derivative (ref models.py:65-73 + autograd):
    du1 (bf16), dW2 / db2 accumulated into the bucket; ragged row counts, both model widths, the headline row count."""
    from owl_vit_object_detection_amd import _lib
    u1 = rnd(rows, D, seed=4).bfloat16(); h1 = torch.nn.functional.gelu(u1.float()).bfloat16()
    w2 = rnd(4, D, scale=0.1, seed=1); b2 = rnd(4, seed=2)
    dboxes = rnd(rows, 4, seed=5)
    # f32 reference through autograd (h1 taken as the bf16 tensor the forward saved)
    u = u1.float().requires_grad_(True); W = w2.clone().requires_grad_(True); bb = b2.clone().requires_grad_(True)
    hq = h1.float() + (torch.nn.functional.gelu(u) - torch.nn.functional.gelu(u).detach())     # value = saved bf16 h1, gradient = gelu'(u)
    s = torch.sigmoid(hq @ W.t() + bb)
    boxes = torch.stack([s[:, 0] - 0.5 * s[:, 2], s[:, 1] - 0.5 * s[:, 3], s[:, 0] + 0.5 * s[:, 2], s[:, 1] + 0.5 * s[:, 3]], -1)
    boxes.backward(dboxes)
    sig = s.detach().contiguous()
    du1 = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
    nblk = _lib.load().owl_box_final_bwd_blocks(rows)
    part = torch.zeros(nblk, 5 * D + 4, device=DEV)
    g = torch.ones(4 * D + 4, device=DEV)                    # dW2 followed by db2, as in the flat bucket; accumulated onto
    cs = torch.ones(D, device=DEV)                           # += column sums of du1 (dense1's bias gradient)
    ops.box_final_bwd(dboxes, sig, h1, u1, w2, du1, part, g, rows, D, du1_colsum=cs)
    report("du1", du1.float(), u.grad, 1e-3, 4e-3)
    report("dW2", g[:4 * D].view(4, D) - 1.0, W.grad, 1e-2 * float(W.grad.abs().max()), 1e-3)
    report("db2", g[4 * D:] - 1.0, bb.grad, 1e-3 * max(1.0, float(bb.grad.abs().max())), 1e-3)
    ref_cs = u.grad.sum(0)
    report("du1 column sums", cs - 1.0, ref_cs, 1e-3 * float(ref_cs.abs().max()) + 1e-4, 1e-3)
    g2 = torch.ones(4 * D + 4, device=DEV); du2 = torch.zeros_like(du1); cs2 = torch.ones(D, device=DEV)
    ops.box_final_bwd(dboxes, sig, h1, u1, w2, du2, part, g2, rows, D, du1_colsum=cs2)
    assert torch.equal(du1, du2) and torch.equal(g, g2) and torch.equal(cs, cs2)     # fixed-order partial sums


def test_cast_and_transpose():
    x = rnd(1000, 37)
    y = ops.cast_bf16(x.contiguous())
    assert torch.equal(y, x.bfloat16())
    a = rnd(200, 136).bfloat16()
    t = torch.zeros(136, 200, dtype=torch.bfloat16, device=DEV)
    ops.transpose_bf16(a, t, 200, 136)
    assert torch.equal(t, a.t().contiguous())


@pytest.mark.parametrize("R,C", [(304, 136), (2312, 768), (200, 64), (1000, 3072)])
def test_transpose_colsum(R, C):
    a = ops.zeros_rows(R, C, torch.bfloat16, DEV); a[:R] = rnd(R, C).bfloat16()
    Rp = ops.pad_rows(R)
    t = torch.zeros(C, Rp, dtype=torch.bfloat16, device=DEV)
    cs = torch.ones(C, device=DEV)
    ops.transpose_colsum(a, t, cs, R, C)
    assert torch.equal(t[:, :R], a[:R].t())
    assert float(t[:, R:].abs().max()) == 0.0 if Rp > R else True
    report("colsum", cs, 1.0 + a[:R].float().sum(0), 2e-2, 1e-3)
    cs2 = torch.zeros(C, device=DEV)
    ops.transpose_colsum(a, None, cs2, R, C)          # column sums only
    report("colsum only", cs2, a[:R].float().sum(0), 2e-2, 1e-3)


@pytest.mark.parametrize("rows,n_out,n_in,splits", [(73984, 768, 768, 28), (2312, 3072, 768, 9), (4624, 768, 3072, 7), (1000, 256, 512, 3),
                                                      (64, 256, 256, 1), (73728, 512, 768, 42), (73728, 32, 512, 128), (1000, 8, 256, 4), (300, 40, 512, 2)])
def test_gemm_tn_slab_weight_gradient(rows, n_out, n_in, splits):
    """dW = dY^T X straight from the token-major operands (LDS transpose-reads) vs an f64 reference on a row sample and
    vs the explicit-transpose NT path; any `rows` (the last K-tile is zero-filled); deterministic."""
    torch.manual_seed(rows + n_out)
    dy = torch.zeros(ops.pad_rows(rows), n_out, device=DEV, dtype=torch.bfloat16); dy[:rows] = (torch.randn(rows, n_out, device=DEV) * 0.1).bfloat16()
    x = torch.zeros(ops.pad_rows(rows), n_in, device=DEV, dtype=torch.bfloat16); x[:rows] = torch.randn(rows, n_in, device=DEV).bfloat16()
    dy[rows:] = 7.0; x[rows:] = 3.0                     # pad rows must NOT leak into the sum
    slab = torch.zeros(splits * n_out * n_in, device=DEV)
    ns = ops.gemm_tn_slab(dy, x, slab, rows, n_out, n_in, splits)
    assert 1 <= ns <= splits
    got = slab[: ns * n_out * n_in].view(ns, n_out, n_in).sum(0)
    ref = dy[:rows].float().T @ x[:rows].float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-3 * scale + 1e-3, (err, scale)          # f32 accumulation, different summation order
    slab2 = torch.zeros_like(slab)
    assert ops.gemm_tn_slab(dy, x, slab2, rows, n_out, n_in, splits) == ns and torch.equal(slab, slab2)     # bitwise repeatable
    # the ping-pong schedule (variant 2 = what variant 0 picks) and the single-phase kernel (variant 1) accumulate in the same order: same bits
    for variant in (1, 2, 2, 1):
        slab3 = torch.zeros_like(slab)
        assert ops.gemm_tn_slab(dy, x, slab3, rows, n_out, n_in, splits, variant=variant) == ns
        assert torch.equal(slab3[: ns * n_out * n_in], slab[: ns * n_out * n_in]), variant
    cs = torch.zeros(n_out, device=DEV)
    ops.colsum_bf16(dy, cs, rows, n_out)
    assert (cs - dy[:rows].float().sum(0)).abs().max().item() <= 1e-3 * rows ** 0.5 + 1e-3
    cs2 = torch.zeros(n_out, device=DEV)
    ops.colsum_bf16(dy, cs2, rows, n_out)
    assert torch.equal(cs, cs2)                         # bitwise repeatable
    # round 6: the column sums of dY (the bias gradient) out of the SAME pass -- one partial per split and feature, from the MFMA fragments (v_dot2c against (1, 1)):
    # the weight slabs keep their bits, the partials add up to the column sums (every split's token range, pad rows and the zero-filled tail excluded)
    from owl_vit_object_detection_amd import _lib
    slab4 = torch.zeros_like(slab); bslab = torch.full((splits, n_out), 9.0, device=DEV)
    assert ops.gemm_tn_slab(dy, x, slab4, rows, n_out, n_in, splits, bias_slab=bslab) == ns
    assert torch.equal(slab4[: ns * n_out * n_in], slab[: ns * n_out * n_in])
    ref_b = dy[:rows].double().sum(0)
    assert (bslab[:ns].double().sum(0) - ref_b).abs().max().item() <= 1e-3 * rows ** 0.5 + 1e-3
    assert bool((bslab[ns:] == 9.0).all())
    per = ((rows + 63) // 64 + ns - 1) // ns * 64          # tokens per split (K-tiles of 64)
    for s_ in (0, ns - 1):
        part = dy[s_ * per: min(rows, (s_ + 1) * per)].double().sum(0)
        assert (bslab[s_].double() - part).abs().max().item() <= 1e-3 * per ** 0.5 + 1e-3, s_
    db = torch.ones(n_out, device=DEV)
    _lib.call("owl_slab_reduce", ops.stream(), bslab, db, n_out, n_out, ns, 1)
    assert (db.double() - 1.0 - ref_b).abs().max().item() <= 1e-3 * rows ** 0.5 + 1e-3
    bslab2 = torch.zeros_like(bslab)
    ops.gemm_tn_slab(dy, x, slab4, rows, n_out, n_in, splits, bias_slab=bslab2)
    assert torch.equal(bslab2[:ns], bslab[:ns])            # bitwise repeatable
    with pytest.raises(_lib.OwlLibError, match="ping-pong kernel only"):
        ops.gemm_tn_slab(dy, x, slab4, rows, n_out, n_in, splits, variant=1, bias_slab=bslab2)


@pytest.mark.parametrize("R,C", [(73984, 768), (1000, 512), (37, 4), (2312, 1024)])
def test_colsum_f32(R, C):
    from owl_vit_object_detection_amd import _lib
    torch.manual_seed(R)
    x = torch.randn(R, C, device=DEV)
    out = torch.full((C,), 0.5, device=DEV)
    ops.colsum_f32(x, out, R, C)
    ref = 0.5 + x.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 1e-4 * R ** 0.5 + 1e-4
    out2 = torch.full((C,), 0.5, device=DEV)
    ops.colsum_f32(x, out2, R, C)
    assert torch.equal(out, out2)                       # fixed-order partial sums, no atomics: bitwise repeatable


def test_add2_layernorm_matches_two_separate_adds():
    """(x + d1) + d2 formed inside the LayerNorm equals storing x + d1 first and adding d2 in the next call, bit for bit;
    store_x=False leaves x untouched."""
    torch.manual_seed(5)
    rows, D = 1000, 768
    x = torch.randn(ops.pad_rows(rows), D, device=DEV)
    d1 = (torch.randn(ops.pad_rows(rows), D, device=DEV) * 0.3).bfloat16()
    d2 = (torch.randn(ops.pad_rows(rows), D, device=DEV) * 0.3).bfloat16()
    g = torch.randn(D, device=DEV); b = torch.randn(D, device=DEV)
    # reference: two stored adds
    xa = x.clone(); ha = torch.zeros_like(d1); hb = torch.zeros_like(d1)
    ops.layernorm(xa, g, b, ha, rows, D, delta=d1)                       # xa = x + d1
    ops.layernorm(xa, g, b, hb, rows, D, delta=d2)                       # xa = (x + d1) + d2, hb = LN(xa)
    # deferred: first LN without the store, second one adds both
    xb = x.clone(); h1 = torch.zeros_like(d1); h2 = torch.zeros_like(d1); xo = torch.zeros_like(x)
    ops.layernorm(xb, g, b, h1, rows, D, delta=d1, store_x=False)
    assert torch.equal(xb, x)                                            # not stored
    assert torch.equal(h1, ha)
    ops.layernorm(xb, g, b, h2, rows, D, delta=d1, delta2=d2, x_out=xo)
    assert torch.equal(h2, hb) and torch.equal(xo[:rows], xa[:rows])
    # torch reference of the sum
    ref = (x[:rows] + d1[:rows].float()) + d2[:rows].float()
    assert torch.equal(xo[:rows], ref)


@pytest.mark.parametrize("B,H,T", [(2, 3, 333), (1, 2, 37), (2, 12, 2305), (1, 16, 3601), (3, 12, 577)])
def test_attention_fwd_vrow_matches_vt_variant_bitwise(B, H, T):
    """The library default (variant 0) is the plain tiling (variant 1) or the peeled one, selected by T alone.  Tuning builds also carry the round-1
    form that takes V^T per head: V read row-major through the LDS transpose-reads gives its very bits (same MFMAs, same order)."""
    tuning = _TUNING_BUILD()
    torch.manual_seed(B * 100 + T)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    vt = torch.zeros(B * D * Tp + 256, device=DEV, dtype=torch.bfloat16)
    vt[: B * D * Tp].view(B, D, Tp)[:] = qkv[:M, 2 * D:].reshape(B, Tp, D).transpose(1, 2)
    o1 = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); l1 = torch.zeros(B, H, Tp, device=DEV)
    o2 = torch.zeros_like(o1); l2 = torch.zeros_like(l1)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o2, D, l2, B, H, T, Tp, 0.125, variant=1)
    if tuning:
        ops.attention_fwd(qkv, qkv[:, D:], 3 * D, vt, D * Tp, o1, D, l1, B, H, T, Tp, 0.125)
        assert torch.equal(o1, o2) and torch.equal(l1, l2)
    assert bool(torch.isfinite(o2.float()).all())
    o3 = torch.zeros_like(o1); l3 = torch.zeros_like(l1)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o3, D, l3, B, H, T, Tp, 0.125, variant=0)
    peeled = (T - 1) % 64 == 0 and T >= 65
    if not peeled:
        assert torch.equal(o3, o2) and torch.equal(l3, l2)
    else:       # other summation order: same values to bf16 / f32 round-off, and the pad rows stay untouched
        assert (o3.float() - o2.float()).abs().max().item() < 2e-2 and (l3 - l2).abs().max().item() < 4e-3
        assert not torch.equal(o3, o2)


def _vrow_reference(qkv, B, H, T, Tp):
    D = H * 64
    v = qkv[: B * Tp].view(B, Tp, 3, H, 64)[:, :T].float()
    q, k, vv = (v[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(2, 3) * 0.125
    return (torch.softmax(sc, -1) @ vv).permute(0, 2, 1, 3).reshape(B, T, D), torch.logsumexp(sc, -1) / math.log(2.0)


@pytest.mark.parametrize("B,H,T", [(2, 2, 65), (3, 1, 129), (2, 3, 193), (1, 2, 257), (2, 12, 577), (2, 12, 2305), (1, 3, 3585), (1, 1, 8193)])
def test_attention_fwd_peeled_class_token(B, H, T):
    """Peeled tiling (variant 2; what the model's T = 1 + patches runs): token 0 enters as the initial softmax state of every other
    query and is itself one VALU-only workgroup per (image, head).  Against f32 softmax, every block-count / idle-wave shape (T - 1 =
    64 .. 95 x 64), output and LSE, with and without LSE; pad rows untouched; repeatable bits."""
    torch.manual_seed(T)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = torch.randn(B, T, 3 * D, device=DEV).bfloat16()
    want, lse_want = _vrow_reference(qkv, B, H, T, Tp)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV); lse = torch.zeros(B, H, Tp, device=DEV)
    out[:] = 7.0; lse[:] = 7.0
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=2)
    report(f"peeled attn T={T}", out[:M].view(B, Tp, D)[:, :T], want, 2e-2, 2e-2)
    report(f"peeled attn T={T}, class-token row", out[:M].view(B, Tp, D)[:, 0], want[:, 0], 1e-2, 1e-2)
    report(f"peeled lse T={T}", lse[:, :, :T], lse_want, 2e-3, 1e-3)
    assert bool((out[:M].view(B, Tp, D)[:, T:] == 7.0).all()) and bool((lse[:, :, T:] == 7.0).all())
    out2 = torch.zeros_like(out)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out2, D, None, B, H, T, Tp, 0.125, variant=2)
    assert torch.equal(out2[:M].view(B, Tp, D)[:, :T], out[:M].view(B, Tp, D)[:, :T])
    # what variant 0 resolves to at these T
    out3 = torch.zeros_like(out)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out3, D, None, B, H, T, Tp, 0.125)
    assert torch.equal(out3, out2)


def test_attention_fwd_peeled_rejects_other_lengths():
    B, H, T = 1, 1, 300
    Tp = 304; D = 64; M = Tp
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV); out = ops.zeros_rows(M, D, torch.bfloat16, DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, variant=2)


@pytest.mark.parametrize("spike_key,spike_q,gain", [(250, 10, 12.0), (0, 5, 12.0), (0, 0, 12.0), (2304, 2000, 12.0), (70, 0, 12.0), (0, 700, -12.0), (1, 1, 12.0)])
def test_attention_fwd_peeled_offset_paths(spike_key, spike_q, gain):
    """Scores far outside the exp2 range of an offset-free sweep: a spike at key 0 makes s0 the initial offset of that query (positive: every
    later tile rides on the offset MFMA; negative: the first tile must rescale the initial state away), a spike in a later tile takes the
    slow path with an initial state to rescale; query 0 is the VALU workgroup (explicit maximum).  All against f32 softmax."""
    B, H, T = 1, 2, 2305
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    x = rnd(B, T, 3 * D, seed=spike_key + 7 * spike_q)
    x[0, spike_key, D:D + 64] = gain * torch.sign(x[0, spike_q, :64] + 1e-3)       # head 0: q . k ~ gain * |q|_1
    x[0, spike_q, :64] *= 6.0
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = x.bfloat16()
    want, lse_want = _vrow_reference(qkv, B, H, T, Tp)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=2)
    assert bool(torch.isfinite(out[:M].float()).all()) and bool(torch.isfinite(lse[:, :, :T]).all())
    report("peeled attn, spiked", out[:M].view(B, Tp, D)[:, :T], want, 3e-2, 2e-2)
    # scores of several hundred: the bf16 rounding of the pre-scaled Q (2^-9 per element, both tilings alike) moves the LSE by ~0.1 against
    # f32 -- the tight check is against the plain tiling, which shares that rounding
    report("peeled lse, spiked, vs f32", lse[:, :, :T], lse_want, 0.5, 2e-3)
    out1 = torch.zeros_like(out); lse1 = torch.zeros_like(lse)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out1, D, lse1, B, H, T, Tp, 0.125, variant=1)
    report("peeled lse, spiked, vs plain tiling", lse[:, :, 1:T], lse1[:, :, 1:T], 5e-3, 1e-5)
    report("peeled attn, spiked, vs plain tiling", out[:M].view(B, Tp, D)[:, 1:T], out1[:M].view(B, Tp, D)[:, 1:T], 1e-2, 1e-2)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("above_log2,expect_slow", [(60.0, False), (80.0, False), (95.0, True), (120.0, True)])
def test_attention_fwd_verdict_threshold_at_kernel_level(variant, above_log2, expect_slow):
    """the stale-offset verdict sits at 2^88 (attention_fwd_common.h) since round 4 -- the spiked-score cases above were sized for the old 2^40 and
    jump straight to scores of several hundred.  Here ONE key of a later tile scores `above_log2` (log2 domain) above the offset the wave holds, for every query of
    head 0: 60 and 80 must ride the fast path (slow_tiles == 0: P up to 2^80 beside O accumulators with 2^48 of f32 headroom left) and still match the f32 softmax;
    95 and 120 must take the slow path (slow_tiles > 0) and match too.  Plain tiling and the class-token-peeled one."""
    B, H, T = 1, 2, 577
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    x = rnd(B, T, 3 * D, seed=int(above_log2) + variant) * 0.05            # base scores ~ 1e-2: the held offset stays 0
    q0 = 8.0
    x[0, :, 0] = q0                                                        # head 0, feature 0 of every query
    x[0, :, D] = 0.0                                                       # ... of every key: 0, but for the one spiking key (tile 4 of the sweep)
    x[0, 250, D] = above_log2 / (q0 * 0.125 * 1.4426950408889634)
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = x.bfloat16()
    want, lse_want = _vrow_reference(qkv, B, H, T, Tp)
    s_spike = float(qkv[250, D].float()) * q0 * 0.125 * 1.4426950408889634
    assert abs(s_spike - above_log2) < 0.5                                  # (after the bf16 rounding of the key)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV); lse = torch.zeros(B, H, Tp, device=DEV)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=variant, slow_tiles=counter)
    torch.cuda.synchronize()
    slow = int(counter.item())
    print(f"variant {variant}: spike {s_spike:.1f} (log2) above the held offset -> slow-path tiles {slow}")
    assert bool(torch.isfinite(out[:M].float()).all()) and bool(torch.isfinite(lse[:, :, :T]).all())
    report(f"attn, one key 2^{above_log2:.0f} above the offset", out[:M].view(B, Tp, D)[:, :T], want, 2e-2, 2e-2)
    report("lse", lse[:, :, :T], lse_want, 5e-2, 2e-3)
    assert (slow > 0) == expect_slow, (slow, above_log2)


@pytest.mark.parametrize("B,H,T", [(2, 3, 333), (1, 2, 37), (1, 12, 577), (1, 12, 2305), (2, 4, 2305), (1, 16, 3601)])
def test_attention_bwd_matches_torch_autograd(B, H, T):
    """dQ / dK / dV of the fused backward (every transposed operand read by the LDS transpose hardware) against f32 autograd of
    softmax(QK^T/8)V on the same bf16 inputs; bf16 outputs -> 2e-2 of the largest gradient."""
    torch.manual_seed(B + T)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16)
    qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125)
    do = torch.zeros_like(o)
    do[:M].view(B, Tp, D)[:, :T] = (torch.randn(B, T, D, device=DEV) * 0.1).bfloat16()
    dvec = torch.zeros(B, H, Tp, device=DEV); dqkv = torch.zeros_like(qkv)
    ops.attention_bwd(qkv, do, o, lse, dvec, dqkv, B, H, T, Tp, 0.125)
    x = qkv[:M].view(B, Tp, 3, H, 64)[:, :T].float()
    q = x[:, :, 0].permute(0, 2, 1, 3).clone().requires_grad_(True)
    k = x[:, :, 1].permute(0, 2, 1, 3).clone().requires_grad_(True)
    v = x[:, :, 2].permute(0, 2, 1, 3).clone().requires_grad_(True)
    ref = torch.softmax((q @ k.transpose(2, 3)) * 0.125, -1) @ v
    ref.backward(do[:M].view(B, Tp, H, 64)[:, :T].permute(0, 2, 1, 3).float())
    got = dqkv[:M].view(B, Tp, 3, H, 64)[:, :T].float()
    for i, g in enumerate((q.grad, k.grad, v.grad)):
        d = (got[:, :, i].permute(0, 2, 1, 3) - g).abs().max().item()
        assert d < 2e-2 * max(g.abs().max().item(), 1e-3), (i, d, g.abs().max().item())
    assert float(dqkv[:M].view(B, Tp, 3 * D)[:, T:].abs().max()) == 0.0 if Tp > T else True      # pad tokens get no gradient


@pytest.mark.parametrize("rows,D,bf16_dy", [(1000, 768, True), (333, 1024, True), (257, 128, False)])
def test_layernorm_bwd_matches_torch_autograd(rows, D, bf16_dy):
    """owl_layernorm_bwd vs torch autograd of F.layer_norm: dx (+ residual gradient), its bf16 copy, dgamma / dbeta (accumulated into the
    caller's buffers through fixed-order partial sums) and -- fused for the trainable layer's LN2 -- the column sums of dx."""
    torch.manual_seed(rows)
    x = torch.randn(ops.pad_rows(rows), D, device=DEV) * 2 + 0.3
    gamma = torch.randn(D, device=DEV); beta = torch.randn(D, device=DEV)
    dy = torch.randn(ops.pad_rows(rows), D, device=DEV) * 0.1
    if bf16_dy:
        dy = dy.bfloat16()
    dres = torch.randn(ops.pad_rows(rows), D, device=DEV) * 0.05
    h = torch.zeros(ops.pad_rows(rows), D, device=DEV, dtype=torch.bfloat16); stats = torch.zeros(ops.pad_rows(rows), 2, device=DEV)
    ops.layernorm(x, gamma, beta, h, rows, D, stats)
    xr = x[:rows].clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy[:rows].float())
    want_dx = xr.grad + dres[:rows]
    for fused in (False, True) if bf16_dy else (False,):
        dx = torch.zeros_like(x); dxb = torch.zeros_like(h)
        dg = torch.full((D,), 0.25, device=DEV); db = torch.full((D,), -0.5, device=DEV); dcs = torch.full((D,), 2.0, device=DEV)
        ops.layernorm_bwd(dy, x, stats, gamma, dres, dx, dg, db, rows, D, dx_bf16=dxb, dx_colsum=dcs if fused else None)
        report("ln_bwd dx", dx[:rows], want_dx, 1e-4, 1e-4)
        assert torch.equal(dxb[:rows], dx[:rows].bfloat16())
        report("ln_bwd dgamma", dg - 0.25, gr.grad, 2e-3, 1e-3)
        report("ln_bwd dbeta", db + 0.5, br.grad, 2e-3, 1e-3)
        if fused:
            report("ln_bwd colsum(dx)", dcs - 2.0, dx[:rows].double().sum(0).float(), 2e-3, 1e-4)
            dcs2 = torch.full((D,), 2.0, device=DEV); dg2 = torch.full((D,), 0.25, device=DEV); db2 = torch.full((D,), -0.5, device=DEV)
            ops.layernorm_bwd(dy, x, stats, gamma, dres, dx, dg2, db2, rows, D, dx_bf16=dxb, dx_colsum=dcs2)
            assert torch.equal(dcs, dcs2) and torch.equal(dg, dg2) and torch.equal(db, db2)        # fixed-order reduction: repeatable bits
    # dx-only form (frozen layers): no parameter gradients, no scratch
    dx = torch.zeros_like(x)
    ops.layernorm_bwd(dy, x, stats, gamma, None, dx, None, None, rows, D)
    report("ln_bwd dx only", dx[:rows], xr.grad, 1e-4, 1e-4)


@pytest.mark.skipif(_TUNING, reason="the shipped library's refusals; a tuning build carries the experiments")
def test_shipped_library_refuses_the_tuning_only_kernels():
    """the default libowlhip.so exports one entry per fused op; experiments are refused loudly, not silently re-routed."""
    from owl_vit_object_detection_amd import _lib
    B, H, T, Tp, D = 1, 1, 193, 200, 64
    qkv = ops.zeros_rows(Tp, 3 * D, torch.bfloat16, DEV); out = ops.zeros_rows(Tp, D, torch.bfloat16, DEV)
    with pytest.raises(RuntimeError, match="OWL_TUNING"):
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, variant=3)
    with pytest.raises(_lib.OwlLibError, match="variant must be 0"):
        _lib.call("owl_attention_fwd_vrow_bf16", ops.stream(), qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, 3, None)
    A = ops.zeros_rows(512, 128, torch.bfloat16, DEV); W = torch.zeros(256, 128, dtype=torch.bfloat16, device=DEV); o = ops.zeros_rows(512, 256, torch.bfloat16, DEV)
    for tile in (8, 9, 5, 4):
        with pytest.raises(_lib.OwlLibError, match="tile must be 0"):
            ops.gemm(ops.EPI_BIAS_BF16, A, W, o, M=512, tile=tile)
    for epi in (ops.EPI_ATOMIC_F32, ops.EPI_TRANS_BF16):
        with pytest.raises(_lib.OwlLibError, match="OWL_TUNING build"):
            ops.gemm(epi, A, W, torch.zeros(512, 256, device=DEV), M=512, Tp=8)
    lib = _lib.load()
    for sym in ("owl_attention_fwd_bf16", "owl_attention_fwd_w64_bf16", "owl_attention_fwd_workspace_bytes", "owl_gemm_fr_ablate", "owl_gemm_set_persistent"):
        assert not hasattr(lib, sym), sym


@pytest.mark.parametrize("ns,n,stride", [(1152, 3076, 3844), (1152, 768, 3844), (130, 32, 64), (127, 3076, 3844), (36, 2359296, 2359296)])
def test_slab_reduce_tall_and_wide(ns, n, stride):
    """owl_slab_reduce: out (+)= sum over `ns` partial rows.  Many partials x few columns (box_final_bwd's 1152 workgroup partials) take the tall kernel (round 6:
    8 lanes per column quad, fixed-order combination), everything else the one-thread-per-quad kernel; both against an f64 sum, with and without accumulation,
    repeatable bits, columns beyond n untouched."""
    from owl_vit_object_detection_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(ns + n)
    slabs = torch.randn(ns, stride, generator=g).to(DEV)
    ref = slabs[:, :n].double().sum(0)
    out = torch.full((stride,), 7.0, device=DEV)
    _lib.call("owl_slab_reduce", ops.stream(), slabs, out, n, stride, ns, 0)
    assert float((out[:n].double() - ref).abs().max()) < 2e-4 * max(1.0, ns ** 0.5 / 8)
    assert bool((out[n:] == 7.0).all())
    out2 = torch.full((stride,), 7.0, device=DEV)
    _lib.call("owl_slab_reduce", ops.stream(), slabs, out2, n, stride, ns, 0)
    assert torch.equal(out, out2)
    acc = torch.ones(stride, device=DEV)
    _lib.call("owl_slab_reduce", ops.stream(), slabs, acc, n, stride, ns, 1)
    assert float((acc[:n].double() - 1.0 - ref).abs().max()) < 2e-4 * max(1.0, ns ** 0.5 / 8)
