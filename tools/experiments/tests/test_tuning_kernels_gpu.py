"""Tuning-build experiments (OWL_TUNING=1 build of libowlhip.so: tools/experiments/build.sh) -- moved out of tests/ in round 6 so that the
product's suite holds no test that the shipped library must skip.  Run: OWL_TUNING=1 python -m pytest tools/experiments/tests -q -m gpu"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from owl_vit_object_detection_amd import ops  # noqa: E402
from tests.test_kernels_gpu import DEV, _TUNING_BUILD, _vrow_reference, gemm_tile, report, rnd  # noqa: E402,F401


@pytest.mark.skipif(not _TUNING_BUILD(), reason="tuning-build experiment (OWL_TUNING=1 build of libowlhip.so): not in the shipped ABI")
def test_gemm_transposed_epilogue(gemm_tile):
    B, Tp, T, K, N = 2, 152, 150, 128, 192          # 3 heads of 64
    M = B * Tp
    A = ops.zeros_rows(M, K, torch.bfloat16, DEV); A[:M] = rnd(M, K).bfloat16()
    W = rnd(N, K, scale=0.1, seed=1).bfloat16(); bias = rnd(N, seed=2)
    out = torch.zeros(B, N, Tp, dtype=torch.bfloat16, device=DEV)
    ops.gemm(ops.EPI_TRANS_BF16, A, W, out, bias=bias, M=M, Tp=Tp)
    ref = (A[:M].float() @ W.float().t() + bias).view(B, Tp, N).permute(0, 2, 1)
    report("trans", out, ref, 2e-2, 1e-2)


# ---- attention forward, one wave per SIMD (variant 3; csrc/attention_fwd_w64.hip) -----------------------------------------------------------------
@pytest.mark.skipif(not _TUNING_BUILD(), reason="tuning-build experiment (OWL_TUNING=1 build of libowlhip.so): not in the shipped ABI")
@pytest.mark.parametrize("B,H,T", [(1, 1, 193), (1, 1, 257), (2, 3, 577), (3, 2, 1025), (2, 12, 2305), (1, 2, 449), (1, 1, 3585)])
def test_attention_fwd_one_wave_per_simd(B, H, T):
    """64 queries per wave, softmax interleaved with the neighbouring tiles' MFMAs (peeled tiling, T - 1 = 3 .. 56 key tiles, odd and even, full and
    partly idle query blocks): output and LSE against f32 softmax; same O bits as the classic peeled kernel (same MFMA chains and order -- only the
    row sums are added in another order); pad rows untouched; repeatable bits; no block flagged on ordinary scores."""
    torch.manual_seed(T + B)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = torch.randn(B, T, 3 * D, device=DEV).bfloat16()
    want, lse_want = _vrow_reference(qkv, B, H, T, Tp)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV); lse = torch.zeros(B, H, Tp, device=DEV)
    out[:] = 7.0; lse[:] = 7.0
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=3)
    torch.cuda.synchronize()
    nq = (T - 1 + 255) // 256
    assert int(ops.attention_redo_ws(B, H, T, out.device)[: B * H * nq].sum()) == 0
    report(f"w64 attn T={T}", out[:M].view(B, Tp, D)[:, :T], want, 2e-2, 2e-2)
    report(f"w64 lse T={T}", lse[:, :, :T], lse_want, 2e-3, 1e-3)
    assert bool((out[:M].view(B, Tp, D)[:, T:] == 7.0).all()) and bool((lse[:, :, T:] == 7.0).all())
    o2 = torch.zeros_like(out); l2 = torch.zeros_like(lse)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o2, D, l2, B, H, T, Tp, 0.125, variant=2)
    v = lambda t: t[:M].view(B, Tp, D)[:, :T].float()
    assert (v(out) - v(o2)).abs().max().item() <= 1.6e-2 and (lse[:, :, :T] - l2[:, :, :T]).abs().max().item() < 1e-4      # (one bf16 ulp where 1 / l rounds apart)
    assert float((v(out) != v(o2)).float().mean()) < 2e-3
    o3 = torch.zeros_like(out)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o3, D, None, B, H, T, Tp, 0.125, variant=3)
    assert torch.equal(v(o3), v(out))


@pytest.mark.skipif(not _TUNING_BUILD(), reason="tuning-build experiment (OWL_TUNING=1 build of libowlhip.so): not in the shipped ABI")
@pytest.mark.parametrize("spike_key,spike_q,gain", [(250, 10, 12.0), (0, 5, 12.0), (2304, 2000, 12.0), (70, 0, 12.0), (0, 700, -12.0), (1, 1, 12.0)])
def test_attention_fwd_one_wave_per_simd_redo_path(spike_key, spike_q, gain):
    """Scores outside the range of the offset-free softmax: the block raises its flag and the classic kernel redoes exactly that block in the same
    call -- its bits (explicit-maximum slow path), the other blocks keep the fast kernel's."""
    B, H, T = 1, 2, 2305
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    x = rnd(B, T, 3 * D, seed=spike_key + 7 * spike_q)
    x[0, spike_key, D:D + 64] = gain * torch.sign(x[0, spike_q, :64] + 1e-3)
    x[0, spike_q, :64] *= 6.0
    qkv = ops.zeros_rows(M, 3 * D, torch.bfloat16, DEV)
    qkv[:M].view(B, Tp, 3 * D)[:, :T] = x.bfloat16()
    want, lse_want = _vrow_reference(qkv, B, H, T, Tp)
    out = ops.zeros_rows(M, D, torch.bfloat16, DEV); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, lse, B, H, T, Tp, 0.125, variant=3)
    torch.cuda.synchronize()
    flags = ops.attention_redo_ws(B, H, T, out.device)[: B * H * 9].view(H, 9).clone()
    assert bool(torch.isfinite(out[:M].float()).all()) and bool(torch.isfinite(lse[:, :, :T]).all())
    report("w64 attn, spiked", out[:M].view(B, Tp, D)[:, :T], want, 3e-2, 2e-2)
    o2 = torch.zeros_like(out); l2 = torch.zeros_like(lse)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o2, D, l2, B, H, T, Tp, 0.125, variant=2)
    if spike_q >= 1:            # (query 0 is the class-token row: the VALU workgroup of both variants, never flagged)
        blk = (spike_q - 1) // 256
        assert int(flags[0, blk]) == 1 or spike_key == 0        # a spike at key 0 flags through the class-token score check
    assert int(flags[1].sum()) == 0                              # head 1 has ordinary scores
    for h in range(H):
        for b9 in range(9):
            if int(flags[h, b9]):       # redone blocks: the classic kernel's very bits
                rows = slice(1 + b9 * 256, 1 + (b9 + 1) * 256)
                assert torch.equal(out[:M].view(B, Tp, D)[0, rows, h * 64:(h + 1) * 64], o2[:M].view(B, Tp, D)[0, rows, h * 64:(h + 1) * 64])
                assert torch.equal(lse[0, h, rows], l2[0, h, rows])


@pytest.mark.skipif(not _TUNING_BUILD(), reason="tuning-build experiment (OWL_TUNING=1 build of libowlhip.so): not in the shipped ABI")
def test_attention_fwd_one_wave_per_simd_rejects_other_lengths():
    B, H, T = 1, 1, 129
    Tp = 136; D = 64
    qkv = ops.zeros_rows(Tp, 3 * D, torch.bfloat16, DEV); out = ops.zeros_rows(Tp, D, torch.bfloat16, DEV)
    with pytest.raises(RuntimeError, match="one wave per SIMD"):
        ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, D, None, B, H, T, Tp, 0.125, variant=3)


