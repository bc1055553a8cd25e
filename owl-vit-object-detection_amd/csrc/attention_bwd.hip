// Fused self-attention backward (autograd form of HF5:377-402) for the trainable encoder layer.
// Flash-style: probabilities are recomputed from the saved log-sum-exp, never read from HBM.
//   D[q]   = sum_d dO[q,d] O[q,d]
//   P      = exp2(S*c - LSE[q]),  S = Q K^T,  c = scale*log2(e)
//   dV     = P^T dO ;  dP = dO V^T ;  dA = P * (dP - D[q]) ;  dQ = scale * dA K ;  dK = scale * dA^T Q
// Two kernels, no atomics:
//   * dkdv: a workgroup owns 128 keys (4 waves x 32) and streams 64-query tiles; per wave
//       S[q,k], dP[q,k] (A = Q / dO rows from LDS, B = K / V fragments held in registers),
//       dV^T[d,k] += dO^T[d,q] P[q,k] and dK^T[d,k] += Q^T[d,q] dA[q,k]: the accumulator registers of
//       the first pair ARE the B operands of the second pair (same trick as the forward kernel);
//   * dq: a workgroup owns 128 queries and streams 64-key tiles; S^T[k,q], dP^T[k,q], then
//       dQ^T[d,q] += K^T[d,k] dA^T[k,q].
// Operands whose contraction index must be the fast per-lane index come from the per-head transposed
// copies (Q^T, K^T, dO^T as [B][heads*64][Tp]) written by the GEMM's transposing epilogue.
// All tiles go HBM -> LDS by LDS-DMA, double-buffered, XOR-swizzled as in gemm.hip.
#include "common.h"
#include <type_traits>

struct AttnBwdP {
    const bf16_t* qkv; int64_t ld_qkv;       // row-major [B*Tp, 3D]: q | k | v
    const bf16_t* dO; int64_t ld_do;         // row-major [B*Tp, D]
    const bf16_t* O;                         // row-major [B*Tp, D]
    const float* lse; float* dvec;           // [B][H][Tp]
    bf16_t* dqkv;                            // row-major [B*Tp, 3D]
    int B, T, Tp, H, D;
    float scale, scale_log2e;
};

__device__ __forceinline__ int swap23b(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }
__device__ __forceinline__ int tile_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int tile_off_v(int row, int ch) { return row * 128 + ((ch ^ swz_vrow(row)) << 4); }   // transpose-friendly image

// ---- D vector ------------------------------------------------------------------------------------
// ---- VALU-lean recomputation (same idea as attention_fwd.hip: these kernels are bound by VALU issue, not by the matrix pipe) ----
// P = exp2(s*c - lse) and dS = P * (dP - D) need, per score, an fma and a subtract whose second operands are constants of
// the score's QUERY.  Both are moved into the matrix pipe: the operand that is loaded once per kernel (this lane's K row in
// the dK/dV kernel, its Q row in the dQ kernel) is pre-scaled by c, and one extra MFMA per accumulator adds the per-query
// constant: contraction slots 0 and 1 carry (hi, lo) = a two-term bf16 split of -lse (resp. -D) on the query side and
// (1, 1) on the key side, so the sum hi + lo enters the f32 accumulator with ~2^-17 relative error.
__device__ __forceinline__ unsigned split_hi_lo_bf16(float x) {          // packed (hi | lo << 16), hi + lo ~= x
    const float hi = bf2f(f2bf(x));
    return pack_bf2(hi, x - hi);
}
__device__ __forceinline__ bf16x8 frag_slot01(unsigned w0, int hi_half) {  // 8-element K-fragment with slots 0,1 = w0 (first half-wave only)
    const uint4 u = make_uint4(hi_half == 0 ? w0 : 0u, 0u, 0u, 0u);
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 f, float c) {
    unsigned wq[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const unsigned u = ((const unsigned*)&f)[e];
        wq[e] = pack_bf2(__uint_as_float(u << 16) * c, __uint_as_float(u & 0xffff0000u) * c);
    }
    const uint4 q4 = make_uint4(wq[0], wq[1], wq[2], wq[3]);
    return __builtin_bit_cast(bf16x8, q4);
}

__global__ __launch_bounds__(256) void attn_dvec_kernel(AttnBwdP p) {
    // D[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d].  One thread per 16-byte chunk (8 elements) of a token row, consecutive
    // threads on consecutive chunks (fully coalesced; the first version gave a thread a whole 128-byte head segment and
    // over-fetched 7x by FETCH_SIZE); the 8 chunks of a head are 8 adjacent lanes -> three xor-shuffles.
    const int cpr = p.D / 8;                                            // chunks per row
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over B*Tp*cpr (cpr is a multiple of 8)
    const int64_t row = i / cpr;
    const int ci = (int)(i - row * cpr);
    float acc = 0.f;
    const bool ok = row < (int64_t)p.B * p.Tp;
    const int64_t b = ok ? row / p.Tp : 0;
    const int t = ok ? (int)(row - b * p.Tp) : 0;
    if (ok && t < p.T) {
        const us8 av = *(const us8*)(p.dO + row * p.ld_do + ci * 8), ov = *(const us8*)(p.O + row * p.ld_do + ci * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) acc += bf2f(av[e]) * bf2f(ov[e]);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (ok && (ci & 7) == 0) p.dvec[((int64_t)b * p.H + (ci >> 3)) * p.Tp + t] = acc;
}

// ---- dK, dV ------------------------------------------------------------------------------------------
static constexpr int BWD1_STAGE = 2 * 8192 + 512;   // Q, dO tiles (row-major; the Q^T / dO^T fragments come out of them by LDS transpose-reads) + lse[64] + dvec[64]

__global__ __launch_bounds__(256, 3) void attn_bwd_dkdv_kernel(AttnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    // XCD-aware 1-D grid (workgroups id, id+8, ... share an XCD / L2): every key block of one (image, head) runs on the
    // same XCD, so the Q / dO / Q^T / dO^T tiles it streams (1.2 MB per pair at T = 2305) are fetched into ONE L2 instead
    // of all eight (FETCH_SIZE of the first version: 6x the algorithmic bytes)
    const int nblk = (p.T + 127) / 128;
    const int pair = ((int)(blockIdx.x >> 3) / nblk) * 8 + (int)(blockIdx.x & 7);
    if (pair >= p.B * p.H) return;
    const int blk = (int)(blockIdx.x >> 3) % nblk;
    const int h = pair % p.H, b = pair / p.H;
    const int k0 = blk * 128 + w * 32;
    const float c = p.scale_log2e;
    const int D = p.D;

    int key = k0 + l31;
    const bool key_ok = key < p.T;
    if (!key_ok) key = p.T - 1;
    const bf16_t* krow = p.qkv + ((int64_t)b * p.Tp + key) * p.ld_qkv + D + h * 64;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        kf[kc] = *(const bf16x8*)(krow + kc * 16 + hi * 8);
        vf[kc] = *(const bf16x8*)(krow + D + kc * 16 + hi * 8);
    }

    bf16x8 ks[4];                                                 // c * K: S arrives in the exp2 domain
#pragma unroll
    for (int kc = 0; kc < 4; kc++) ks[kc] = scale_frag(kf[kc], c);
    const bf16x8 ones01 = frag_slot01(0x3F803F80u, hi);            // key side: 1.0 in contraction slots 0, 1
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const bf16_t* qbase = p.qkv + (int64_t)b * p.Tp * p.ld_qkv + h * 64;
    const bf16_t* dobase = p.dO + (int64_t)b * p.Tp * p.ld_do + h * 64;
    const float* lsebase = p.lse + ((int64_t)b * p.H + h) * p.Tp;
    const float* dvbase = p.dvec + ((int64_t)b * p.H + h) * p.Tp;

    // Staging.  Full tiles go through `buffer_load_dwordx4 ... offen lds`: the (image, head) bases sit in buffer
    // descriptors, the tile offset in an SGPR and the lane's row/chunk offsets in VGPRs computed once -- no address VALU
    // per tile (the global_load_lds form needs a 64-bit VGPR address per piece).  Only the partial last tile clamps rows.
    // Both tiles are row-major [64 query][64 d] in the transpose-friendly image (common.h): the dV / dK MFMAs' dO^T / Q^T operands
    // are read out of them by ds_read_b64_tr_b16 -- no transposed copies in HBM, half the LDS-DMA pieces per tile.
    // A wave's two pieces of a tile are rows w*8 + (lane>>3) and 32 further: the swizzle has period 16 rows, so both share ONE
    // lane offset per operand and the 32-row step rides in the scalar offset (two live VGPRs instead of four across the loop).
    unsigned row_voff[2];                                          // {Q, dO}
    {
        const int r = w * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swz_vrow(r);
        row_voff[0] = (unsigned)((r * p.ld_qkv + ch * 8) * 2);
        row_voff[1] = (unsigned)((r * p.ld_do + ch * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t do_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, 0x7fffffff, 0x00020000);
    const int q_tile_bytes = (int)(64 * p.ld_qkv * 2), do_tile_bytes = (int)(64 * p.ld_do * 2);
    auto stage_full = [&](int buf, int qt) {
        unsigned char* base = lds + buf * BWD1_STAGE;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, LPTR(base + r0 * 128), 16, (int)row_voff[0], qt * q_tile_bytes + qd * (q_tile_bytes >> 1), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, LPTR(base + 8192 + r0 * 128), 16, (int)row_voff[1], qt * do_tile_bytes + qd * (do_tile_bytes >> 1), 0, 0);
        }
    };
    auto stage_clamped = [&](int buf, int qt) {                    // partial last tile: rows >= T re-read row T-1 (32-bit offsets only)
        unsigned char* base = lds + buf * BWD1_STAGE;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            const int r = r0 + (lane >> 3);
            const int ch = (lane & 7) ^ swz_vrow(r);
            int q = qt * 64 + r;
            if (q >= p.T) q = p.T - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, LPTR(base + r0 * 128), 16, (q * (int)p.ld_qkv + ch * 8) * 2, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, LPTR(base + 8192 + r0 * 128), 16, (q * (int)p.ld_do + ch * 8) * 2, 0, 0, 0);
        }
    };
    auto stage = [&](int buf, int qt) {                            // wave-uniform choice
        if (qt * 64 + 64 <= p.T) stage_full(buf, qt); else stage_clamped(buf, qt);
    };
    // lse / D values of a tile travel through a register: loaded when the tile is staged, written to
    // LDS after the compute phase (keeps ordinary loads out of the LDS-DMA window)
    auto load_scal = [&](int qt) -> float {
        const int i = threadIdx.x & 63;
        const int q = qt * 64 + i;
        if (threadIdx.x < 64) return (q < p.T) ? lsebase[q] : 1e30f;       // finite "infinity": it goes through a bf16 hi/lo split
        if (threadIdx.x < 128) return (q < p.T) ? dvbase[q] : 0.f;
        return 0.f;
    };
    auto store_scal = [&](int buf, float v) {
        // stored already negated and split into (hi, lo) bf16: the consumers use the word as contraction slots 0, 1 directly
        if (threadIdx.x < 128) ((unsigned*)(lds + buf * BWD1_STAGE + 16384))[threadIdx.x] = split_hi_lo_bf16(-v);
    };

    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    unsigned tr_off[2][2];                                        // [d-block][half of the lane's 8 queries]
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) tr_off[d][h2] = tr_lane_off(lane, d, h2);
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) { dv[d][r] = 0.f; dk[d][r] = 0.f; }

    const int nq = (p.T + 63) / 64;
    stage(0, 0);
    store_scal(0, load_scal(0));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int qt = 0; qt < nq; qt++) {
        float pend = 0.f;
        if (qt + 1 < nq) { stage(cur ^ 1, qt + 1); pend = load_scal(qt + 1); }
        const unsigned char* tb = lds + cur * BWD1_STAGE;
        const unsigned* lse_t = (const unsigned*)(tb + 16384);   // packed (hi, lo) of -lse / -D per query of the tile
        const unsigned* dv_t = lse_t + 64;
        // a wave whose 32 keys all lie beyond T (T = 2305: three of the last block's four waves) only stages and syncs
#pragma unroll
        for (int sub = 0; sub < (k0 < p.T ? 2 : 0); sub++) {
            const int qrow = sub * 32 + swap23b(l31);
            // query-side fragments of the two constant MFMAs: this lane's A row is query `qrow` of the tile
            const bf16x8 a_lse = frag_slot01(lse_t[qrow], hi);
            const bf16x8 a_dv = frag_slot01(dv_t[qrow], hi);
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lse, ones01, zero16, 0, 0, 0);    // -lse[query] in every key column
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_dv, ones01, zero16, 0, 0, 0);    // -D[query]
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const bf16x8 qa = *(const bf16x8*)(tb + tile_off_v(qrow, kc * 2 + hi));
                const bf16x8 da = *(const bf16x8*)(tb + 8192 + tile_off_v(qrow, kc * 2 + hi));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, ks[kc], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kc], dp, 0, 0, 0);
            }
            // register r <-> query sub*32 + 16*(r>>3) + 8*hi + (r&7)
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                float pv[8], dsv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    pv[j] = __builtin_amdgcn_exp2f(s[cc * 8 + j]);            // = exp2(score*c - lse)
                    dsv[j] = pv[j] * dp[cc * 8 + j];                           // P * (dP - D)
                }
                const uint4 pw = make_uint4(pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3]), pack_bf2(pv[4], pv[5]), pack_bf2(pv[6], pv[7]));
                const uint4 dsw = make_uint4(pack_bf2(dsv[0], dsv[1]), pack_bf2(dsv[2], dsv[3]), pack_bf2(dsv[4], dsv[5]), pack_bf2(dsv[6], dsv[7]));
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dsw);
                const unsigned tq = lds0 + cur * BWD1_STAGE + (sub * 2 + cc) * 2048;      // 16 queries further = 2 KiB
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const bf16x8 dot = lds_tr8(tq + 8192 + tr_off[d][0], tq + 8192 + tr_off[d][1]);   // dO^T[d*32 + lane&31][8 queries]
                    const bf16x8 qt_ = lds_tr8(tq + tr_off[d][0], tq + tr_off[d][1]);                 // Q^T
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt_, dsf, dk[d], 0, 0, 0);
                }
            }
        }
        if (qt + 1 < nq) store_scal(cur ^ 1, pend);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    uint4 stk[2][2], stv[2][2];
    pack_token_rows(dk, p.scale, stk);              // 16-byte stores (common.h)
    pack_token_rows(dv, 1.0f, stv);
    // (output addresses formed after the loop from opaque scalars: computed in the prologue they are spilled across the whole loop)
    int k0e = k0, be = b, he = h;
    asm volatile("" : "+s"(k0e), "+s"(be), "+s"(he));
    const int key_e = k0e + l31;
    if (key_e < p.T) {
        bf16_t* orow = p.dqkv + ((int64_t)be * p.Tp + key_e) * p.ld_qkv + he * 64;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const int dd = d * 32 + 16 * pr + 8 * hi;
                *(uint4*)(orow + D + dd) = stk[d][pr];
                *(uint4*)(orow + 2 * D + dd) = stv[d][pr];
            }
    }
}

// ---- dQ ----------------------------------------------------------------------------------------------
static constexpr int BWD2_STAGE = 2 * 8192;   // K, V tiles (row-major; K^T fragments come out of the K tile by LDS transpose-reads)

__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int nblk = (p.T + 127) / 128;                          // XCD-aware 1-D grid, as in the dK/dV kernel
    const int pair = ((int)(blockIdx.x >> 3) / nblk) * 8 + (int)(blockIdx.x & 7);
    if (pair >= p.B * p.H) return;
    const int blk = (int)(blockIdx.x >> 3) % nblk;
    const int h = pair % p.H, b = pair / p.H;
    const int q0 = blk * 128 + w * 32;
    const float c = p.scale_log2e;
    const int D = p.D;

    int q = q0 + l31;
    const bool q_ok = q < p.T;
    if (!q_ok) q = p.T - 1;
    const bf16_t* qrow = p.qkv + ((int64_t)b * p.Tp + q) * p.ld_qkv + h * 64;
    const bf16_t* dorow = p.dO + ((int64_t)b * p.Tp + q) * p.ld_do + h * 64;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        qf[kc] = *(const bf16x8*)(qrow + kc * 16 + hi * 8);
        dof[kc] = *(const bf16x8*)(dorow + kc * 16 + hi * 8);
    }
    const float lse_q = p.lse[((int64_t)b * p.H + h) * p.Tp + q];
    const float dvec_q = p.dvec[((int64_t)b * p.H + h) * p.Tp + q];
    bf16x8 qs[4];                                                 // c * Q: S^T arrives in the exp2 domain
#pragma unroll
    for (int kc = 0; kc < 4; kc++) qs[kc] = scale_frag(qf[kc], c);
    const bf16x8 ones01 = frag_slot01(0x3F803F80u, hi);            // key side: 1.0 in contraction slots 0, 1
    const bf16x8 q_lse = frag_slot01(split_hi_lo_bf16(-lse_q), hi);   // query side: -lse  (hi, lo)
    const bf16x8 q_dv = frag_slot01(split_hi_lo_bf16(-dvec_q), hi);   //             -D
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const bf16_t* kbase = p.qkv + (int64_t)b * p.Tp * p.ld_qkv + D + h * 64;
    // staging: buffer loads with scalar tile offsets for full tiles, clamped rows for the last one.  Both tiles are row-major
    // [64 key][64 d] with the transpose-friendly chunk swizzle (common.h): the dQ MFMA's K^T operand is read out of the K tile by
    // ds_read_b64_tr_b16 -- no K^T copy in HBM, a third less LDS-DMA per tile.
    unsigned row_voff;                                             // one lane offset for both pieces and both tiles (see the dK/dV kernel)
    {
        const int r = w * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swz_vrow(r);
        row_voff = (unsigned)((r * p.ld_qkv + ch * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(kbase + D), 0, 0x7fffffff, 0x00020000);
    const int k_tile_bytes = (int)(64 * p.ld_qkv * 2);
    auto stage_full = [&](int buf, int kv) {
        unsigned char* base = lds + buf * BWD2_STAGE;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, (int)row_voff, kv * k_tile_bytes + qd * (k_tile_bytes >> 1), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + 8192 + r0 * 128), 16, (int)row_voff, kv * k_tile_bytes + qd * (k_tile_bytes >> 1), 0, 0);
        }
    };
    auto stage_clamped = [&](int buf, int kv) {
        unsigned char* base = lds + buf * BWD2_STAGE;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            const int r = r0 + (lane >> 3);
            const int ch = (lane & 7) ^ swz_vrow(r);
            int key = kv * 64 + r;
            if (key >= p.T) key = p.T - 1;
            const int voff = (key * (int)p.ld_qkv + ch * 8) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, voff, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + 8192 + r0 * 128), 16, voff, 0, 0, 0);
        }
    };
    auto stage = [&](int buf, int kv) {
        if (kv * 64 + 64 <= p.T) stage_full(buf, kv); else stage_clamped(buf, kv);
    };

    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    unsigned tr_off[2][2];                                        // [d-block][half of the lane's 8 keys]
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) tr_off[d][h2] = tr_lane_off(lane, d, h2);
    f32x16 dq[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) dq[d][r] = 0.f;

    const int nkv = (p.T + 63) / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // one 64-key tile.  TAIL (compile time) only for the peeled partial last tile: in the full tiles the per-element
    // key mask (compare + select per score) is not even compiled in -- it used to cost ~4 VALU per element of every tile.
    auto tile = [&](int cur, int kv, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const unsigned char* tb = lds + cur * BWD2_STAGE;
        // idle waves (all 32 queries beyond T) only stage and sync; a tail tile whose keys all sit in its first half skips the second
        const int nsub = (q0 >= p.T) ? 0 : ((TAIL && p.T - kv * 64 <= 32) ? 1 : 2);
#pragma unroll
        for (int sub = 0; sub < nsub; sub++) {
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones01, q_lse, zero16, 0, 0, 0);    // -lse of the lane's query
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones01, q_dv, zero16, 0, 0, 0);    // -D
            const int krow = sub * 32 + swap23b(l31);
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const bf16x8 ka = *(const bf16x8*)(tb + tile_off_v(krow, kc * 2 + hi));
                const bf16x8 va = *(const bf16x8*)(tb + 8192 + tile_off_v(krow, kc * 2 + hi));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qs[kc], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[kc], dp, 0, 0, 0);
            }
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                float dsv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float pv = __builtin_amdgcn_exp2f(s[cc * 8 + j]);          // = exp2(score*c - lse)
                    if constexpr (TAIL) {
                        const int key = kv * 64 + sub * 32 + cc * 16 + 8 * hi + j;
                        if (key >= p.T) pv = 0.f;
                    }
                    dsv[j] = pv * dp[cc * 8 + j];                              // P * (dP - D)
                }
                const uint4 dsw = make_uint4(pack_bf2(dsv[0], dsv[1]), pack_bf2(dsv[2], dsv[3]), pack_bf2(dsv[4], dsv[5]), pack_bf2(dsv[6], dsv[7]));
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, dsw);
                const unsigned tk = lds0 + cur * BWD2_STAGE + (sub * 2 + cc) * 2048;     // 16 keys further = 2 KiB
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const bf16x8 kt = lds_tr8(tk + tr_off[d][0], tk + tr_off[d][1]);       // K^T[d*32 + lane&31][8 keys]
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, dsf, dq[d], 0, 0, 0);
                }
            }
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int nfull = p.T / 64;                                   // tiles with no key >= T
    int cur = 0;
    for (int kv = 0; kv < nfull; kv++) {
        if (kv + 1 < nkv) stage(cur ^ 1, kv + 1);
        tile(cur, kv, std::false_type{});
        sync();
        cur ^= 1;
    }
    if (nfull < nkv) tile(cur, nfull, std::true_type{});
    uint4 stq[2][2];
    pack_token_rows(dq, p.scale, stq);              // 16-byte stores (common.h)
    int q0e = q0, be = b, he = h;                   // (see the dK/dV epilogue)
    asm volatile("" : "+s"(q0e), "+s"(be), "+s"(he));
    const int q_e = q0e + l31;
    if (q_e < p.T) {
        bf16_t* orow = p.dqkv + ((int64_t)be * p.Tp + q_e) * p.ld_qkv + he * 64;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int pr = 0; pr < 2; pr++) *(uint4*)(orow + d * 32 + 16 * pr + 8 * hi) = stq[d][pr];
    }
}

OWL_API int owl_attention_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tp, int64_t* bytes) {
    OWL_CHECK_ARG(bytes && B > 0 && H > 0 && Tp > 0, "owl_attention_bwd_workspace_bytes: bad arguments");
    *bytes = B * H * Tp * (int64_t)sizeof(float);          // dvec_ws: one f32 per (image, head, token)
    return 0;
}

OWL_API int owl_attention_bwd_bf16(void* stream, const void* qkv, const void* dO, const void* O,
                                      const float* lse, float* dvec_ws, void* dqkv, int64_t B, int64_t H, int64_t T, int64_t Tp,
                                      float scale, int phases) {
    OWL_CHECK_ARG(qkv && dO && O && lse && dvec_ws && dqkv, "owl_attention_bwd_bf16: null pointer");
    OWL_CHECK_ARG(phases >= 0 && phases <= 7, "owl_attention_bwd_bf16: phases is a mask of 1 (dvec) | 2 (dK, dV) | 4 (dQ); 0 = all");
    if (phases == 0) phases = 7;
    OWL_CHECK_ARG(Tp % 8 == 0 && T > 0 && T <= Tp, "owl_attention_bwd_bf16: Tp %% 8, T <= Tp");
    AttnBwdP p{};
    p.D = (int)(H * 64);
    p.qkv = (const bf16_t*)qkv; p.ld_qkv = 3 * (int64_t)p.D;
    p.dO = (const bf16_t*)dO; p.ld_do = p.D; p.O = (const bf16_t*)O;
    p.lse = lse; p.dvec = dvec_ws; p.dqkv = (bf16_t*)dqkv;
    p.B = (int)B; p.T = (int)T; p.Tp = (int)Tp; p.H = (int)H;
    p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BWD1_STAGE);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BWD2_STAGE);
    });
    const int64_t nd = B * Tp * (H * 8);                       // one thread per 16-byte chunk
    if (phases & 1) {
        hipLaunchKernelGGL(attn_dvec_kernel, dim3((unsigned)((nd + 255) / 256), 1, 1), dim3(256), 0, s, p);
        OWL_LAUNCH_CHECK();
    }
    const int64_t npairs8 = (B * H + 7) / 8;                  // (image, head) pairs per XCD, rounded up
    dim3 grid((unsigned)(npairs8 * ((T + 127) / 128) * 8));
    // dK / dV and dQ only share their inputs (and write disjoint column thirds of dqkv): a caller may give them to two streams behind the dvec pass (`phases`)
    if (phases & 2) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), 2 * BWD1_STAGE, s, p);
        OWL_LAUNCH_CHECK();
    }
    if (phases & 4) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 2 * BWD2_STAGE, s, p);
        OWL_LAUNCH_CHECK();
    }
    return 0;
}
