// Shared pieces of the bf16 MFMA GEMM family: epilogue ids, kernel parameters, per-accumulator-tile
// epilogues (used identically by the 128x128 and the 256x256 block-tile kernels).
#pragma once
#include "common.h"

// MI355X (gfx950) is the only target: 256 compute units in 8 XCDs.  The persistent grids, the remainder-round rule and the small-problem (tile 6) rule of
// csrc/gemm.hip count on it; ops.CHIP_CUS is the Python side's copy (models.OwlViT warns if the device reports another count).
constexpr int NUM_CUS = 256;

enum {
    EPI_BIAS_BF16 = 0,   // out bf16 = acc + bias
    EPI_QGELU_BF16 = 1,  // u = acc + bias; out bf16 = u*sigmoid(1.702u); aux (optional) bf16 = quick_gelu'(u) (round 6; before: u) -- what EPI_DQGELU_BF16 multiplies by
    EPI_GELU_BF16 = 2,   // erf GELU, aux (optional) bf16 = u
    EPI_RESID_F32 = 3,   // out f32 = resid + acc + bias
    EPI_F32 = 4,         // out f32 = alpha*acc (+ bias)
    EPI_ATOMIC_F32 = 5,  // atomicAdd(out f32, alpha*acc)            (split-K)
    EPI_TRANS_BF16 = 6,  // out_t[b][n][t] bf16 = acc + bias, m = b*Tp + t   (per-head transposed)
    EPI_PATCH_F32 = 7,   // A gathered from image patches (any patch size >= 8: rows padded to 2^n in the K index only); out f32 [b*Tp + 1 + p][n] = acc + pos[1+p][n]
    EPI_DQGELU_BF16 = 8, // out bf16 = acc * aux   (aux = the quick_gelu' the forward epilogue saved: one multiply, no transcendental)
    EPI_DGELU_BF16 = 9,  // out bf16 = acc * gelu_erf'(aux u)
    EPI_ACC_F32 = 10,    // out f32 += acc   (resid == out)
    EPI_SLAB_F32 = 11,   // split-K partial: out f32 [split][M][N] = alpha*acc   (reduced by owl_slab_reduce)
    EPI_PATCHM_F32 = 12, // as EPI_PATCH_F32 but A is an explicit im2row matrix in the same padded-row K order (the single-phase REFERENCE kernels for patch sizes that are not 2^n, e.g. L/14)
};

struct GemmP {
    const bf16_t* A; int64_t lda; int64_t a_rows;
    const bf16_t* W; int64_t ldw; int64_t w_rows;
    const float* bias;
    void* out; int64_t ldo;
    const float* resid;
    void* aux; int64_t ld_aux;
    int64_t M, N, K;       // M,N: store guards; K multiple of 64
    int tiles_m, tiles_n, kt_per_split, nsplit, persistent;
    float alpha;
    int64_t Tp;            // EPI_TRANS: rows per image
    int64_t P, G, ps, S;   // EPI_PATCH: patches / grid / patch size / image side
    int ps_log2;           // log2 of the PADDED patch-row length psp (= ps for power-of-two patch sizes)
    int ps_magic;          // 65536 / ps + 1: R / ps == (R * ps_magic) >> 16 for the gather's row index R < 3 ps + 8 (patch sizes that are not 2^n)
    const float* pos;      // [T, N]
    int64_t slab_stride;   // EPI_SLAB: elements per split slab
    int stagger;           // gemm_pp2: start delay of every second workgroup of an XCD, in units of s_sleep 127 (0 = none)
    int dbg;               // read by the kernels of an OWL_TUNING build only: bit 1 = stores wrapped into a cache-resident window, bit 2 = non-temporal bf16 stores
};

__device__ __forceinline__ float sigmoid1702_f(float u) {   // 1 / (1 + exp(-1.702 u)) with v_exp / v_rcp
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
}
__device__ __forceinline__ float qgelu_f(float u) { return u * sigmoid1702_f(u); }
// quick_gelu'(u) = s (1 + 1.702 u (1 - s)) from the sigmoid s the forward epilogue has in hand anyway.  Round 6: the quick-GELU epilogue SAVES THIS (bf16) instead
// of the pre-activation, and the backward epilogue (EPI_DQGELU_BF16) is one multiply: no v_exp / v_rcp in the dX GEMM, same bytes.  Explicit operation order
// (mul, sub, fma, mul): every kernel of the family gives the same bits.
__device__ __forceinline__ float dqgelu_from_s(float u, float s) {
    const float t = 1.702f * u, om = 1.0f - s;
    return s * fmaf(t, om, 1.0f);
}
__device__ __forceinline__ float dqgelu_f(float u) { return dqgelu_from_s(u, sigmoid1702_f(u)); }
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float u) { return dgelu_erf_f(u); }   // (common.h)

// ---- XCD-aware bijective tile remap (blocks b, b+8, ... share an XCD / L2) -------------------------
__device__ __forceinline__ int xcd_remap(int bid, int ntile) {
    const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- element-wise epilogue of ONE (row, 4 consecutive columns) quad -----------------------------------
// v[4] = raw accumulators of output row `orow` (store row, already remapped for PATCH) at columns n..n+3.
template <int EPI>
__device__ __forceinline__ void epi_quad(const GemmP& p, int64_t orow, const float* posrow, int64_t n, float (&v)[4], int split) {
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] *= p.alpha;
    if (p.bias) {
        const float4 b4 = *(const float4*)(p.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if constexpr (EPI == EPI_BIAS_BF16) {
        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
        *(uint2*)((bf16_t*)p.out + orow * p.ldo + n) = o;
    } else if constexpr (EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16) {
        float g[4], sv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if constexpr (EPI == EPI_QGELU_BF16) {
                const float sg = sigmoid1702_f(v[e]);
                g[e] = v[e] * sg; sv[e] = dqgelu_from_s(v[e], sg);          // saved: the derivative
            } else {
                g[e] = gelu_f(v[e]); sv[e] = v[e];                           // saved: the pre-activation
            }
        }
        if (p.aux) {
            uint2 a; a.x = pack_bf2(sv[0], sv[1]); a.y = pack_bf2(sv[2], sv[3]);
            *(uint2*)((bf16_t*)p.aux + orow * p.ld_aux + n) = a;
        }
        uint2 o; o.x = pack_bf2(g[0], g[1]); o.y = pack_bf2(g[2], g[3]);
        *(uint2*)((bf16_t*)p.out + orow * p.ldo + n) = o;
    } else if constexpr (EPI == EPI_DQGELU_BF16 || EPI == EPI_DGELU_BF16) {
        const uint2 a = *(const uint2*)((const bf16_t*)p.aux + orow * p.ld_aux + n);
        const float u[4] = {bf2f(a.x & 0xffff), bf2f(a.x >> 16), bf2f(a.y & 0xffff), bf2f(a.y >> 16)};
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= (EPI == EPI_DQGELU_BF16) ? u[e] : dgelu_f(u[e]);
        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
        *(uint2*)((bf16_t*)p.out + orow * p.ldo + n) = o;
    } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_ACC_F32) {
        const float4 r4 = *(const float4*)(p.resid + orow * p.ldo + n);
        *(float4*)((float*)p.out + orow * p.ldo + n) = make_float4(r4.x + v[0], r4.y + v[1], r4.z + v[2], r4.w + v[3]);
    } else if constexpr (EPI == EPI_F32) {
        *(float4*)((float*)p.out + orow * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_SLAB_F32) {
        *(float4*)((float*)p.out + (int64_t)split * p.slab_stride + orow * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_PATCH_F32 || EPI == EPI_PATCHM_F32) {
        const float4 p4 = *(const float4*)(posrow + n);
        *(float4*)((float*)p.out + orow * p.ldo + n) = make_float4(v[0] + p4.x, v[1] + p4.y, v[2] + p4.z, v[3] + p4.w);
    } else if constexpr (EPI == EPI_ATOMIC_F32) {
        float* o = (float*)p.out + orow * p.ldo + n;
#pragma unroll
        for (int e = 0; e < 4; e++) atomicAdd(o + e, v[e]);
    }
}

// ---- LDS-staged, fully coalesced epilogue of one PASS = two side-by-side 32x32 accumulator tiles --------
// The accumulator layout gives a lane 4 consecutive elements of 16 different rows; stored directly that is 32
// partial cache lines per store instruction (measured: ~40 % of the whole GEMM at K = 768).  Instead the wave
// writes the pass as a [32 rows][64 cols] f32 image into its PRIVATE 8 KiB of LDS (the two 4 KiB pieces it
// itself DMA-fills for the next K-tile -- free after the barrier, no cross-wave hazard), 16-byte chunks
// XOR-swizzled by (row & 15) (conflict-free ds_write_b128 / ds_read_b128), then reads it back row-contiguous:
// one wave-instruction = 4 rows x 256 B, so every global access of the epilogue (stores, residual / aux
// loads) covers whole cache lines.
//   t0 / t1 : the two accumulator tiles (columns 0-31 / 32-63 of the pass image)
//   TRANS = false: image rows = output rows m (row_base + r), image cols = output cols n (col_base + c)
//   TRANS = true : image rows = output cols n (row_base + r), image cols = token rows m (col_base + c)
template <int EPI>
__device__ __forceinline__ void epi_pass(const GemmP& p, const f32x16& t0, const f32x16& t1, unsigned char* pieceA,
                                         unsigned char* pieceB, int64_t row_base, int64_t col_base, int lane, int split) {
    constexpr bool TRANS = (EPI == EPI_TRANS_BF16);
    const int hi = lane >> 5, r = lane & 31;
    unsigned char* wrow = (r < 16 ? pieceA : pieceB) + (r & 15) * 256;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        const int c0 = 2 * qd + hi, c1 = 8 + c0;
        *(float4*)(wrow + ((c0 ^ (r & 15)) << 4)) = make_float4(t0[qd * 4 + 0], t0[qd * 4 + 1], t0[qd * 4 + 2], t0[qd * 4 + 3]);
        *(float4*)(wrow + ((c1 ^ (r & 15)) << 4)) = make_float4(t1[qd * 4 + 0], t1[qd * 4 + 1], t1[qd * 4 + 2], t1[qd * 4 + 3]);
    }
    const int ch = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 4 + (lane >> 4);
        const unsigned char* rrow = (row < 16 ? pieceA : pieceB) + (row & 15) * 256;
        const f32x4 f = lds_read_f4(rrow + ((ch ^ (row & 15)) << 4));
        float v[4] = {f[0], f[1], f[2], f[3]};
        if constexpr (!TRANS) {
            const int64_t m = row_base + row, n = col_base + ch * 4;
            if (m >= p.M || n >= p.N) continue;
            int64_t orow = m;
            const float* posrow = nullptr;
            if constexpr (EPI == EPI_PATCH_F32 || EPI == EPI_PATCHM_F32) {
                const int64_t b = m / p.P, pp = m - b * p.P;
                orow = b * p.Tp + 1 + pp;
                posrow = p.pos + (1 + pp) * p.N;
            }
            epi_quad<EPI>(p, orow, posrow, n, v, split);
        } else {
            const int64_t n = row_base + row, m = col_base + ch * 4;
            if (n >= p.N || m >= p.M) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const int64_t b = m / p.Tp, t = m - b * p.Tp;
            uint2 o;
            o.x = pack_bf2(v[0] + bv, v[1] + bv);
            o.y = pack_bf2(v[2] + bv, v[3] + bv);
            *(uint2*)((bf16_t*)p.out + (b * p.N + n) * p.Tp + t) = o;
        }
    }
}

// ---- register-resident bf16 epilogue of ONE 32x32 accumulator tile (wide stores) -------------------------
// The 32x32 accumulator gives a lane 4 consecutive outputs per register quad and the other half-wave the adjacent 4:
// one v_permlane32_swap per packed word pairs them into 8 consecutive outputs per lane -> 16-byte stores (half the
// store instructions of the natural 8-byte layout).  The two 16-byte chunks per tile are returned to the caller,
// which issues them with epi_store_chunk.
//   non-TRANS (swapped operands): lane owns output row m = m_tile + (lane&31); chunk c covers columns
//       n_tile + 16*c + 8*hi + {0..7}
//   TRANS (natural operands): lane owns output column n = n_tile + (lane&31); chunk c covers token rows
//       m_tile + 16*c + 8*hi + {0..7}
//
// EPI_DQGELU / EPI_DGELU read the saved pre-activation tile (p.aux).  epi_aux_load fetches it in the STORE layout (two 16-byte loads per
// lane and tile instead of four 8-byte ones at a row stride) so that a caller can have the loads of all its tiles in flight before the
// first one is consumed; epi_tile_bf16 then undoes the lane pairing with the same v_permlane32_swap (it is its own inverse).
// The wave's bias values for a whole tile in ONE LDS round trip: bq[j][qd] = floats lds_bias[32 j + 8 qd + 4 hi .. +4] (the columns of accumulator
// quad qd of column tile j).  lds_read_f4 waits for every read on its own -- 32 serialised LDS round trips per wave and tile when the epilogue
// re-reads the slice per 32 x 32 tile and quad (the asm statement cannot be hoisted or merged): ~3000 of the ~4650 ticks of a group's epilogue
// interval at K = 768 (profiles/r03_gemm_anatomy.md section 5).
__device__ __forceinline__ void epi_bias_preload(const float* lds_bias, int hi, f32x4 (&bq)[2][4]) {
    const unsigned a = (unsigned)(uintptr_t)LPTR(lds_bias + 4 * hi);
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
                 "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(bq[0][0]), "=&v"(bq[0][1]), "=&v"(bq[0][2]), "=&v"(bq[0][3]), "=&v"(bq[1][0]), "=&v"(bq[1][1]), "=&v"(bq[1][2]), "=&v"(bq[1][3])
                 : "v"(a) : "memory");
}

template <bool GUARD>
__device__ __forceinline__ void epi_aux_load(const GemmP& p, int64_t m_tile, int64_t n_tile, int lane, uint4 (&a)[2]) {
    const int hi = lane >> 5;
    const int64_t m = m_tile + (lane & 31);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int64_t n = n_tile + 16 * c + 8 * hi;
        a[c] = make_uint4(0u, 0u, 0u, 0u);
        if (!GUARD || (m < p.M && n < p.N)) a[c] = *(const uint4*)((const bf16_t*)p.aux + m * p.ld_aux + n);
    }
}

template <int EPI, bool GUARD>
__device__ __forceinline__ void epi_tile_bf16(const GemmP& p, const f32x16& acc, int64_t m_tile, int64_t n_tile, int lane,
                                              uint4& chunk0, uint4& chunk1, const float* lds_bias, const uint4* aux_pre = nullptr,
                                              const f32x4* bias_pre = nullptr) {      // bias_pre: epi_bias_preload's bq[j] for this column tile
    constexpr bool TRANS = (EPI == EPI_TRANS_BF16);
    const int hi = lane >> 5;
    unsigned w[8];   // packed words, quad qd -> w[2*qd], w[2*qd+1]
    unsigned aw[8] = {};   // the tile saved for / by the backward (p.aux) in the same quad layout: read (activation derivatives) or written (activations)
    if constexpr (TRANS) {
        const int64_t n = n_tile + (lane & 31);
        float bv = 0.f;
        if (p.bias) bv = lds_read_f1(lds_bias + (lane & 31));   // this tile's bias slice, staged in LDS
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            w[2 * qd] = pack_bf2(acc[qd * 4 + 0] + bv, acc[qd * 4 + 1] + bv);
            w[2 * qd + 1] = pack_bf2(acc[qd * 4 + 2] + bv, acc[qd * 4 + 3] + bv);
        }
    } else {
        const int64_t m = m_tile + (lane & 31);
        const bool m_ok = !GUARD || m < p.M;
        const bf16_t* aux_row = (const bf16_t*)p.aux + m * p.ld_aux + n_tile + 4 * hi;
        const float* bias_p = lds_bias + 4 * hi;           // this tile's bias slice, staged in LDS
        if constexpr (EPI == EPI_DQGELU_BF16 || EPI == EPI_DGELU_BF16) {
            if (aux_pre) {
                unsigned ax[4] = {aux_pre[0].x, aux_pre[0].y, aux_pre[1].x, aux_pre[1].y}, ay[4] = {aux_pre[0].z, aux_pre[0].w, aux_pre[1].z, aux_pre[1].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    auto r = __builtin_amdgcn_permlane32_swap(ax[k], ay[k], false, false);
                    ax[k] = r[0]; ay[k] = r[1];
                }
                aw[0] = ax[0]; aw[1] = ax[1]; aw[2] = ay[0]; aw[3] = ay[1]; aw[4] = ax[2]; aw[5] = ax[3]; aw[6] = ay[2]; aw[7] = ay[3];
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            float v[4];
            const bool n_ok = !GUARD || (n_tile + 8 * qd + 4 * hi) < p.N;
            const bool ok = m_ok && n_ok;
            // v = acc * alpha + bias as ONE packed fma per element pair (alpha = 1 in the encoder: acc*1 + b is exactly acc + b)
            const f32x2_t al = {p.alpha, p.alpha};
            f32x2_t b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
            if (p.bias) {                                  // wave-uniform
                const f32x4 b4 = bias_pre ? bias_pre[qd] : lds_read_f4(bias_p + 8 * qd);
                b01 = (f32x2_t){b4[0], b4[1]}; b23 = (f32x2_t){b4[2], b4[3]};
            }
            const f32x2_t v01 = __builtin_elementwise_fma((f32x2_t){acc[qd * 4 + 0], acc[qd * 4 + 1]}, al, b01);
            const f32x2_t v23 = __builtin_elementwise_fma((f32x2_t){acc[qd * 4 + 2], acc[qd * 4 + 3]}, al, b23);
            v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
            if constexpr (EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16) {
                float sv[4] = {v[0], v[1], v[2], v[3]};    // what is saved for the backward: erf-GELU the pre-activation, quick-GELU its derivative (below)
                if constexpr (EPI == EPI_QGELU_BF16) {
                    // u * 1/(1 + exp2(-2.4555 u)), two elements per VALU instruction where the ISA has a packed form
                    // (v_pk_mul_f32 / v_pk_add_f32); v_exp / v_rcp stay scalar.  Same operations as qgelu_f, same rounding.
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2_t u = {v[e], v[e + 1]};
                        const f32x2_t t = u * -2.4554669595930156f;
                        const f32x2_t d = (f32x2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
                        const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                        const f32x2_t o = u * r;
                        if (p.aux) { sv[e] = dqgelu_from_s(u.x, r.x); sv[e + 1] = dqgelu_from_s(u.y, r.y); }     // (wave-uniform branch)
                        v[e] = o.x; v[e + 1] = o.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_f(v[e]);
                }
                if (p.aux) {                               // wave-uniform; saved for the backward (layers the backward passes through only)
                    if constexpr (EPI == EPI_QGELU_BF16) {  // stored below, paired like the output
                        aw[2 * qd] = pack_bf2(sv[0], sv[1]); aw[2 * qd + 1] = pack_bf2(sv[2], sv[3]);
                    } else {
                        uint2 a; a.x = pack_bf2(sv[0], sv[1]); a.y = pack_bf2(sv[2], sv[3]);
                        if (ok) *(uint2*)((bf16_t*)aux_row + 8 * qd) = a;
                    }
                }
            } else if constexpr (EPI == EPI_DQGELU_BF16 || EPI == EPI_DGELU_BF16) {
                uint2 a = make_uint2(0u, 0u);
                if (aux_pre) a = make_uint2(aw[2 * qd], aw[2 * qd + 1]);
                else if (ok) a = *(const uint2*)(aux_row + 8 * qd);
                const float u[4] = {bf2f(a.x & 0xffff), bf2f(a.x >> 16), bf2f(a.y & 0xffff), bf2f(a.y >> 16)};
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] *= (EPI == EPI_DQGELU_BF16) ? u[e] : dgelu_f(u[e]);       // (quick-GELU: the saved derivative itself)
            }
            w[2 * qd] = pack_bf2(v[0], v[1]);
            w[2 * qd + 1] = pack_bf2(v[2], v[3]);
        }
    }
    if constexpr (!TRANS && EPI == EPI_QGELU_BF16) {
        // The tile saved for the backward leaves in the OUTPUT's lane pairing: two 16-byte stores per lane (round 6; it used to leave as four 8-byte
        // stores straight from the accumulator layout: twice the store instructions of the output itself, and behind them the two-phase kernel waited
        // for every acknowledgement -- fc1 of a layer the backward passes through took 497 us against 333).  Same bytes at the same addresses.
        // (erf-GELU, the box head's two 768-wide layers, keeps the direct form: paired it measured 127.5 -> 129.3 us.)
        if (p.aux) {                                       // wave-uniform
            unsigned ax[4] = {aw[0], aw[1], aw[4], aw[5]}, ay[4] = {aw[2], aw[3], aw[6], aw[7]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                auto r = __builtin_amdgcn_permlane32_swap(ax[k], ay[k], false, false);
                ax[k] = r[0]; ay[k] = r[1];
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            const int64_t m = m_tile + (lane & 31);
            bf16_t* arow = (bf16_t*)p.aux + m * p.ld_aux + n_tile + 8 * hi;
            const u32x4_t s0 = {ax[0], ax[1], ay[0], ay[1]}, s1 = {ax[2], ax[3], ay[2], ay[3]};
            if (!GUARD || (m < p.M && n_tile + 8 * hi < p.N)) *(u32x4_t*)arow = s0;
            if (!GUARD || (m < p.M && n_tile + 16 + 8 * hi < p.N)) *(u32x4_t*)(arow + 16) = s1;
        }
    }
    // pair quads (0,1) and (2,3) across the half-waves: X = even quad word, Y = odd quad word;
    // v_permlane32_swap: lanes 32-63 of X <-> lanes 0-31 of Y
    unsigned x[4] = {w[0], w[1], w[4], w[5]}, y[4] = {w[2], w[3], w[6], w[7]};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        auto r = __builtin_amdgcn_permlane32_swap(x[k], y[k], false, false);
        x[k] = r[0]; y[k] = r[1];
    }
    chunk0 = make_uint4(x[0], x[1], y[0], y[1]);
    chunk1 = make_uint4(x[2], x[3], y[2], y[3]);
}

// ---- register-resident bf16 epilogue of ONE ROW BLOCK (32 rows x the wave's 64 columns) with quad-contiguous stores ----------------------
// The CU's store path takes the lanes of a store instruction four at a time: four CONSECUTIVE lanes writing 64 contiguous bytes cost one step,
// four lanes on four different rows four steps (tools/probe/store_pattern.hip: 1.5 us against 4.3 us per 128-KiB tile and CU; pairs 2.9 us; which
// lines an instruction covers does not matter, only what consecutive lanes cover).  The accumulator layout puts a row's data in lanes r and r + 32,
// so epi_tile_bf16's stores are the four-step kind.  Here:
//   * the caller stages the W tile with its rows permuted inside every 64-row group (gemm_pp2.hip, LINES): the accumulator register (j, qd, e) of
//     half-wave hi then holds column 32 hi + 16 j + 4 qd + e of the wave's 64 -- a lane owns 64 CONTIGUOUS bytes of its row (no v_permlane32_swap);
//   * the lane's four 16-byte pieces are transposed against the four lanes of its quad (two butterfly stages of v_cndmask + DPP quad_perm, 16
//     VALU instructions per four words): lane 4k + i then holds piece i of rows 4k .. 4k + 3, and store instruction t writes row 4k + t with the
//     quad's 64 bytes contiguous.
// Same values as epi_tile_bf16 (same operations per element in the same order), same number of store instructions.

__device__ __forceinline__ void epi_lines_bias_preload(const float* lds_bias, int hi, f32x4 (&bq)[8]) {     // floats lds_bias[32 hi + 4 t .. +4], t = 0..7
    const unsigned a = (unsigned)(uintptr_t)LPTR(lds_bias + 32 * hi);
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[4]), "=&v"(bq[5]), "=&v"(bq[6]), "=&v"(bq[7])
                 : "v"(a) : "memory");
}

// 4 x 4 transposition of a lane's four 16-byte pieces d[4t .. 4t+3] against the four lanes of its quad: two butterfly stages; lanes l and l ^ X exchange so
// that the lane with bit X clear ends with (x, partner's x) and the other with (partner's y, y).  hipcc makes a select + v_mov_b32_dpp + two selects of each
// pair and stage.  Its own inverse: "piece t of my row" <-> "piece (lane & 3) of row (lane & 28) + t".
// (Hand-written v_cndmask_b32_dpp pairs -- x' = bit ? partner.y : x, y' = bit ? y : partner.x, half the instructions, VCC set by
// hand -- measured SLOWER in the model, 1209 against 1212.5 img/s on one box: the asm blocks pin the order hipcc otherwise interleaves with the
// conversion of the next row block and the stores.)
__device__ __forceinline__ void quad_transpose16(unsigned (&d)[16], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            unsigned& x = d[8 * g + c]; unsigned& y = d[8 * g + 4 + c];
            const unsigned u = (unsigned)__builtin_amdgcn_mov_dpp((int)(o1 ? x : y), 0xB1, 0xF, 0xF, true);
            if (o1) x = u; else y = u;
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {
            unsigned& x = d[4 * g + c]; unsigned& y = d[4 * g + 8 + c];
            const unsigned u = (unsigned)__builtin_amdgcn_mov_dpp((int)(o2 ? x : y), 0x4E, 0xF, 0xF, true);
            if (o2) x = u; else y = u;
        }
    }
}

// The saved tile of EPI_DQGELU_BF16 (p.aux) for one row block of the quad-contiguous epilogue, LOADED the way that epilogue stores (round 6): instruction t
// reads row (lane & 28) + t, the quad's four lanes 64 contiguous bytes of it -- one step of the CU's load path where epi_aux_load's accumulator-layout pattern
// (a row's two 16-byte pieces in lanes r and r + 32) takes four (tools/probe/store_pattern.hip: 1.5 against 4.3 us per 128-KiB tile and CU).  x[4t .. 4t+3] is
// what instruction t returned; quad_transpose16 turns it into the lane's own 64 bytes (word 8 j + 2 qd + s = columns 16 j + 4 qd + 2 s, + 1 of its half).
template <bool GUARD>
__device__ __forceinline__ void epi_lines_aux_load(const GemmP& p, int64_t m_tile, int64_t n_wave, int lane, unsigned (&x)[16]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const int hi = lane >> 5;
    // wave-uniform base (SGPR pair) + one 32-bit lane offset per load: 64-bit per-lane addresses cost the registers the two row blocks in flight need
    const bf16_t* base = (const bf16_t*)p.aux + m_tile * p.ld_aux + n_wave;
    const unsigned ld = (unsigned)p.ld_aux, col = 32u * hi + 8u * (lane & 3), r0 = (unsigned)(lane & 28);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (!GUARD || (m_tile + r0 + t < p.M && n_wave + col < p.N)) v = *(const u32x4_t*)(base + ((r0 + t) * ld + col));
        x[4 * t] = v[0]; x[4 * t + 1] = v[1]; x[4 * t + 2] = v[2]; x[4 * t + 3] = v[3];
    }
}

template <int EPI, bool GUARD>
__device__ __forceinline__ void epi_lines_bf16(const GemmP& p, const f32x16& acc0, const f32x16& acc1, int64_t m_tile, int64_t n_wave, int lane,
                                               const float* lds_bias, const f32x4* bias_pre = nullptr,   // bias_pre[4 j + qd]: epi_lines_bias_preload
                                               unsigned* aux_words = nullptr) {                         // EPI_DQGELU: the lane's own 64 bytes of p.aux (epi_lines_aux_load + quad_transpose16); overwritten
    static_assert(EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_DQGELU_BF16,
                  "quad-contiguous epilogue: forward bf16 epilogues and the multiply-by-saved-derivative one");
    const int hi = lane >> 5;
    const int64_t m = m_tile + (lane & 31);
    const float* bias_p = lds_bias + 32 * hi;
    unsigned d_own[16], a[16];
    // EPI_DQGELU: the packed products replace the aux words they were formed from (word k of d and of aux_words hold the same two elements): 16 registers
    // less while two row blocks of aux are in flight
    unsigned (&d)[16] = (EPI == EPI_DQGELU_BF16) ? *reinterpret_cast<unsigned (*)[16]>(aux_words) : d_own;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const f32x16& acc = j ? acc1 : acc0;
            float v[4];
            const f32x2_t al = {p.alpha, p.alpha};
            f32x2_t b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
            if (p.bias) {                                  // wave-uniform
                const f32x4 b4 = bias_pre ? bias_pre[4 * j + qd] : lds_read_f4(bias_p + 16 * j + 4 * qd);
                b01 = (f32x2_t){b4[0], b4[1]}; b23 = (f32x2_t){b4[2], b4[3]};
            }
            const f32x2_t v01 = __builtin_elementwise_fma((f32x2_t){acc[qd * 4 + 0], acc[qd * 4 + 1]}, al, b01);
            const f32x2_t v23 = __builtin_elementwise_fma((f32x2_t){acc[qd * 4 + 2], acc[qd * 4 + 3]}, al, b23);
            v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
            if constexpr (EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16) {
                float sv[4] = {v[0], v[1], v[2], v[3]};
                if constexpr (EPI == EPI_QGELU_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2_t u = {v[e], v[e + 1]};
                        const f32x2_t t = u * -2.4554669595930156f;
                        const f32x2_t dd = (f32x2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
                        const f32x2_t r = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
                        const f32x2_t o = u * r;
                        if (p.aux) { sv[e] = dqgelu_from_s(u.x, r.x); sv[e + 1] = dqgelu_from_s(u.y, r.y); }
                        v[e] = o.x; v[e + 1] = o.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_f(v[e]);
                }
                if (p.aux) { a[8 * j + 2 * qd] = pack_bf2(sv[0], sv[1]); a[8 * j + 2 * qd + 1] = pack_bf2(sv[2], sv[3]); }
            } else if constexpr (EPI == EPI_DQGELU_BF16) {
                const unsigned a0 = aux_words[8 * j + 2 * qd], a1 = aux_words[8 * j + 2 * qd + 1];
                v[0] *= bf2f(a0 & 0xffff); v[1] *= bf2f(a0 >> 16); v[2] *= bf2f(a1 & 0xffff); v[3] *= bf2f(a1 >> 16);      // same product as epi_tile_bf16 / epi_quad
            }
            d[8 * j + 2 * qd] = pack_bf2(v[0], v[1]);
            d[8 * j + 2 * qd + 1] = pack_bf2(v[2], v[3]);
        }
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    if constexpr (EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16) {
        if (p.aux) {                                       // wave-uniform; pre-activation save (trainable layer only): the lane's own 64 bytes
            bf16_t* aux_row = (bf16_t*)p.aux + m * p.ld_aux + n_wave + 32 * hi;
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (!GUARD || (m < p.M && n_wave + 32 * hi + 8 * t < p.N))
                    *(u32x4_t*)(aux_row + 8 * t) = (u32x4_t){a[4 * t], a[4 * t + 1], a[4 * t + 2], a[4 * t + 3]};
        }
    }
    quad_transpose16(d, lane);          // piece i of rows 4k .. 4k + 3 into lane 4k + i: store instruction t writes row 4k + t with the quad's 64 bytes contiguous
    const int64_t row0 = m_tile + (lane & 28), n = n_wave + 32 * hi + 8 * (lane & 3);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t row = row0 + t;
        if (GUARD && (row >= p.M || n >= p.N)) continue;
        *(u32x4_t*)((bf16_t*)p.out + row * p.ldo + n) = (u32x4_t){d[4 * t], d[4 * t + 1], d[4 * t + 2], d[4 * t + 3]};
    }
}

// ---- register-resident f32 epilogue of ONE 32x32 accumulator tile (swapped operands: lane owns output row m) ------------
// out f32 = alpha*acc (+ bias)  [EPI_F32]   |   out f32 += alpha*acc  [EPI_ACC_F32]; four 16-byte accesses per lane and tile at columns
// n_tile + 8*qd + 4*hi.  Same operations in the same order as epi_quad (alpha, then bias, then the residual): identical bits to the
// LDS-staged epilogue of the single-phase kernel.
template <int EPI, bool GUARD>
__device__ __forceinline__ void epi_tile_f32(const GemmP& p, const f32x16& acc, int64_t m_tile, int64_t n_tile, int lane, const float* lds_bias,
                                             const f32x4* bias_pre = nullptr) {
    const int hi = lane >> 5;
    const int64_t m = m_tile + (lane & 31);
    const bool m_ok = !GUARD || m < p.M;
    float* orow = (float*)p.out + m * p.ldo + n_tile + 4 * hi;
    const float* bias_p = lds_bias + 4 * hi;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[qd * 4 + e] * p.alpha;
        if (p.bias) {                                  // wave-uniform
            const f32x4 b4 = bias_pre ? bias_pre[qd] : lds_read_f4(bias_p + 8 * qd);
            v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        }
        const bool ok = m_ok && (!GUARD || (n_tile + 8 * qd + 4 * hi) < p.N);
        if (!ok) continue;
        if constexpr (EPI == EPI_ACC_F32) {
            const float4 r4 = *(const float4*)(orow + 8 * qd);
            *(float4*)(orow + 8 * qd) = make_float4(r4.x + v[0], r4.y + v[1], r4.z + v[2], r4.w + v[3]);
        } else {
            *(float4*)(orow + 8 * qd) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ---- register-resident patch-embed epilogue of ONE 32x32 accumulator tile: out f32 [b*Tp + 1 + p][n] = alpha*acc + pos[1 + p][n] for output row
// m = b*P + p (HF5:282-288,336-343; no bias).  Same operations in the same order as epi_quad's patch branch: identical bits to the LDS-staged epilogue.
template <bool GUARD>
__device__ __forceinline__ void epi_tile_patch(const GemmP& p, const f32x16& acc, int64_t m_tile, int64_t n_tile, int lane) {
    const int hi = lane >> 5;
    const int64_t m = m_tile + (lane & 31);
    const bool m_ok = !GUARD || m < p.M;
    const int64_t mm = m_ok ? m : 0;
    const int64_t b = mm / p.P, pp = mm - b * p.P;
    float* orow = (float*)p.out + (b * p.Tp + 1 + pp) * p.ldo + n_tile + 4 * hi;
    const float* prow = p.pos + (1 + pp) * p.N + n_tile + 4 * hi;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        if (!m_ok || (GUARD && (n_tile + 8 * qd + 4 * hi) >= p.N)) continue;
        const float4 p4 = *(const float4*)(prow + 8 * qd);
        *(float4*)(orow + 8 * qd) = make_float4(acc[qd * 4 + 0] * p.alpha + p4.x, acc[qd * 4 + 1] * p.alpha + p4.y,
                                                acc[qd * 4 + 2] * p.alpha + p4.z, acc[qd * 4 + 3] * p.alpha + p4.w);
    }
}

// issue one 16-byte chunk (c = 0/1 within the tile)
template <int EPI, bool GUARD>
__device__ __forceinline__ void epi_store_chunk(const GemmP& p, const uint4& ch, int64_t m_tile, int64_t n_tile, int c, int lane) {
    const int hi = lane >> 5;
    if constexpr (EPI == EPI_TRANS_BF16) {
        const int64_t n = n_tile + (lane & 31), m = m_tile + 16 * c + 8 * hi;
        if (GUARD && (n >= p.N || m >= p.M)) return;
        const int64_t b = m / p.Tp, t = m - b * p.Tp;
        *(uint4*)((bf16_t*)p.out + (b * p.N + n) * p.Tp + t) = ch;
    } else {
        int64_t m = m_tile + (lane & 31), n = n_tile + 16 * c + 8 * hi;
        if (GUARD && (m >= p.M || n >= p.N)) return;
#ifdef OWL_TUNING        // (tuning build only: stores wrapped into a cache-resident window)
        if (p.dbg & 2) { m = (int64_t)(blockIdx.x & 255) * 256 + (m & 255); n &= 255; }
#endif
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        u32x4_t* dst = (u32x4_t*)((bf16_t*)p.out + m * p.ldo + n);
        const u32x4_t val = {ch.x, ch.y, ch.z, ch.w};
#ifdef OWL_TUNING        // (tuning build only: non-temporal output stores -- measured neutral, profiles/r02_gemm_two_phase.md)
        if (p.dbg & 4) { __builtin_nontemporal_store(val, dst); return; }
#endif
        *dst = val;
    }
}
