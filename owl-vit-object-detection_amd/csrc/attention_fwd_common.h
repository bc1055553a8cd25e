// Pieces shared by the attention-forward kernels (attention_fwd.hip: 3 waves per SIMD, 32 queries per wave;
// attention_fwd_w64.hip: one wave per SIMD, 64 queries per wave): launch parameters, the K-row permutation and the class-token row.
#pragma once
#include "common.h"

struct AttnFwdP {
    const bf16_t* q; const bf16_t* k; int64_t ld_qk;   // row-major [B*Tp, ld]; head h at column h*64
    const bf16_t* vt; int64_t vt_img_stride;            // V^T [B][heads..][64][Tp]; element stride per image
    bf16_t* out; int64_t ld_out;                        // [B*Tp, ld_out], head h at column h*64
    float* lse;                                         // optional [B][H][Tp], log2 domain
    int T, Tp, H, B, nqb, dbg;
    float scale_log2e;
    const int* redo;                                    // fix-up mode (behind attention_fwd_w64.hip): run only the 128-query blocks whose
    int redo_nqb;                                       // 256-query block is flagged in redo[pair * redo_nqb + (qb >> 1)]
    int* slow_tiles;                                    // optional device counter: += 1 per (wave, key tile) that leaves the fast path after a wave's first tile
};

// The stale-offset verdict of the forward sweep: a tile keeps the softmax offset its wave already holds as long as every row sum of the tile stays below
// 2^88 (log2 domain: scores up to 88 above the offset = 61 nats); see attention_fwd.hip.  tests/golden/make_golden.py (attention_stats) simulates the
// same rule on the reference's logits to predict the kernel's slow-path count -- keep the two in step.
#define ATTN_VERDICT_LOG2 88.0f
#define ATTN_VERDICT_SUM 3.0948501e26f          // 2^88

__device__ __forceinline__ int swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// VROW = false: V arrives per head TRANSPOSED (V^T [B][heads*64][Tp], written by a transposing GEMM epilogue or a token transpose).
// VROW = true : V is read where the QKV GEMM leaves it (row-major, column 2D + h*64 of the qkv rows); the [64 key][64 d] tile is
//               staged exactly like the K tile and transposed by the LDS hardware (`ds_read_b64_tr_b16`, two per fragment).
// PEEL (VROW only): token 0 (the class token) is taken out of the tiling.  T = 1 + 48^2 = 2305 is one more than 36 key tiles / 18 query
//               blocks: tiled as it is, EVERY query block pays a 37th, masked, one-key tile and every (image, head) a 19th query block
//               with one live query (5.8 % of the launch, tools/attn_peel.py).  Peeled, key 0 enters as the INITIAL STATE of the online
//               softmax (p0 = exp2(s0), l = p0, O = p0 v0: one 64-long dot product per query on the VALU), the tiles cover tokens 1..T-1
//               (36 full tiles, no mask code on the path), and query 0 of every (image, head) is one extra, VALU-only workgroup
//               (attn_cls_row: 2 T dot products of length 64 -- no MFMA tile with 127 dead columns).
//               (Two more sweeps were built and measured in round 2 -- software-pipelined across tiles, "optimistic" without per-tile
//               checks -- and removed again: -5..10 % / +0.5 %, profiles/r02_attn_fwd_experiments.md; the code is in history at 22922f6.)

// Query row 0 of one (image, head) against all T keys, on the VALU.  8 lanes per key row (one 16-byte feature chunk each), so the 256
// threads form 32 row groups; group g runs an online softmax over rows g, g + 32, ... in ONE pass over K and V (5 rows of each per batch -- 6 spilled 5 VGPRs into scratch at the kernel's 168-register budget --,
// the next batch's 10 loads issued before the current one is consumed: the workgroup is L2-latency-bound, nothing else), then the 32
// partial states (m, l, O[64]) are merged through LDS in a fixed order: same bits every launch.
__device__ __forceinline__ void attn_cls_row(const AttnFwdP& p, int b, int h, unsigned char* lds) {
    constexpr int U = 5;
    const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;
    const int T = p.T;
    const int64_t ld = p.ld_qk;
    const bf16_t* kb = p.k + (int64_t)b * p.Tp * ld + h * 64 + sub * 8;
    const bf16_t* vb = p.vt + (int64_t)b * p.Tp * ld + h * 64 + sub * 8;
    float qv[8];
    {
        const uint4 qu = *(const uint4*)(p.q + (int64_t)b * p.Tp * ld + h * 64 + sub * 8);
        const unsigned u[4] = {qu.x, qu.y, qu.z, qu.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { qv[2 * e] = __uint_as_float(u[e] << 16) * p.scale_log2e; qv[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u) * p.scale_log2e; }
    }
    float m = -1e30f, l = 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 kq[2][U], vq[2][U];
    auto fetch = [&](int set, int j0) {          // rows j0 + 32 u (clamped: a row past T is read, not used)
#pragma unroll
        for (int u = 0; u < U; u++) {
            int j = j0 + 32 * u; j = j < T ? j : T - 1;
            kq[set][u] = *(const uint4*)(kb + (int64_t)j * ld);
            vq[set][u] = *(const uint4*)(vb + (int64_t)j * ld);
        }
    };
    auto consume = [&](int set, int j0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned ku[4] = {kq[set][u].x, kq[set][u].y, kq[set][u].z, kq[set][u].w};
            const unsigned vu[4] = {vq[set][u].x, vq[set][u].y, vq[set][u].z, vq[set][u].w};
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) { d += qv[2 * e] * __uint_as_float(ku[e] << 16); d += qv[2 * e + 1] * __uint_as_float(ku[e] & 0xffff0000u); }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if (j0 + 32 * u >= T) d = -1e30f;                    // (every group's first row is a real one: T >= 65)
            const float mn = fmaxf(m, d);
            const float alpha = __builtin_amdgcn_exp2f(m - mn), pj = __builtin_amdgcn_exp2f(d - mn);
            m = mn;
            l = l * alpha + pj;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[2 * e] = acc[2 * e] * alpha + pj * __uint_as_float(vu[e] << 16);
                acc[2 * e + 1] = acc[2 * e + 1] * alpha + pj * __uint_as_float(vu[e] & 0xffff0000u);
            }
        }
    };
    fetch(0, grp);
    int j0 = grp;
    for (; j0 + 32 * U < T; j0 += 64 * U) {      // two batches per trip: the register sets are compile-time
        fetch(1, j0 + 32 * U);
        consume(0, j0);
        if (j0 + 64 * U < T) fetch(0, j0 + 64 * U);
        consume(1, j0 + 32 * U);
    }
    if (j0 < T) consume(0, j0);
    // merge: red[g] = (m, l, O[0..63]) of row group g
    float* red = (float*)lds;
    if (sub == 0) { red[grp * 66] = m; red[grp * 66 + 1] = l; }
#pragma unroll
    for (int e = 0; e < 8; e++) red[grp * 66 + 2 + sub * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        float mm = red[0];
        for (int g = 1; g < 32; g++) mm = fmaxf(mm, red[g * 66]);
        float lt = 0.f, o = 0.f;
        for (int g = 0; g < 32; g++) {
            const float sc = __builtin_amdgcn_exp2f(red[g * 66] - mm);
            lt += red[g * 66 + 1] * sc;
            o += red[g * 66 + 2 + tid] * sc;
        }
        p.out[(int64_t)b * p.Tp * p.ld_out + h * 64 + tid] = f2bf(o / lt);
        if (p.lse && tid == 0) p.lse[((int64_t)b * p.H + h) * p.Tp] = mm + __builtin_amdgcn_logf(lt);
    }
}

