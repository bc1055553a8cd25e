// Fused encoder self-attention forward (flash-style; the [H,T,T] probabilities never reach HBM).
// Reference arithmetic: HF5:377-402 eager_attention_forward as called from HF5:428-459 --
// softmax(Q K^T * dh^-0.5) V, no mask, dropout 0, dh = 64 (B/16, B/32, L/14 alike).
//
// CDNA4 structure: one workgroup = 4 waves = 128 query rows of one (image, head); each wave owns
// 32 query rows.  Scores are computed TRANSPOSED, S^T[key, q] = K Q^T with
// v_mfma_f32_32x32x16_bf16, so that a lane holds 16 keys of ONE query column per 32x32 tile:
//   * row max / row sum are in-lane reductions plus one exchange with lane^32,
//   * the exponentiated accumulator registers ARE the B operand (P^T[key, q]) of the second MFMA,
//     O^T[d, q] += V^T[d, key] P^T[key, q] -- no LDS round trip, no permutes.  K rows are fetched
//     with key bits 2<->3 swapped so that accumulator registers 0..7 / 8..15 are two runs of 8
//     consecutive keys, i.e. exactly one ds_read_b128 of the V^T tile per MFMA.
// K tiles [64 keys][64 d] and V^T tiles [64 d][64 keys] (V^T is written per head by the QKV GEMM's
// transposing epilogue) stream HBM -> LDS by LDS-DMA, double-buffered, XOR-swizzled like the GEMM
// tiles (conflict-free ds_read_b128).  Online softmax in the exp2 domain.
#include "common.h"
#include <type_traits>

struct AttnFwdP {
    const bf16_t* q; const bf16_t* k; int64_t ld_qk;   // row-major [B*Tp, ld]; head h at column h*64
    const bf16_t* vt; int64_t vt_img_stride;            // V^T [B][heads..][64][Tp]; element stride per image
    bf16_t* out; int64_t ld_out;                        // [B*Tp, ld_out], head h at column h*64
    float* lse;                                         // optional [B][H][Tp], log2 domain
    int T, Tp, H, B, nqb, dbg;
    float scale_log2e;
};

__device__ __forceinline__ int swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// VROW = false: V arrives per head TRANSPOSED (V^T [B][heads*64][Tp], written by a transposing GEMM epilogue or a token transpose).
// VROW = true : V is read where the QKV GEMM leaves it (row-major, column 2D + h*64 of the qkv rows); the [64 key][64 d] tile is
//               staged exactly like the K tile and transposed by the LDS hardware (`ds_read_b64_tr_b16`, two per fragment).
// PIPE (VROW only): the software-pipelined sweep described at `pipe sweep` below (2 waves per SIMD instead of 3: it keeps the
//               scores of two tiles and the probabilities of two tiles in registers); SCHED adds explicit issue-order hints.
// OPT (classic structure): "optimistic" sweep -- tile 0 runs the checked tile code, every further tile runs WITHOUT the per-tile row-sum
//               check, offset MFMA and rescale branch (5 VALU + ~20 SALU instructions and two branches per tile); the verdict is taken
//               once, on the accumulated row sums, and a workgroup that fails it redoes its query block with the checked sweep.
template <bool VROW, bool PIPE = false, bool SCHED = false, bool OPT = false>
__global__ __launch_bounds__(256, PIPE ? 2 : 3) void attn_fwd_kernel(AttnFwdP p) {
    // dynamic LDS (one object): with a static array hipcc drains the just-issued LDS-DMA (vmcnt(0)) before the
    // first ds_read of every tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    // XCD-aware work mapping (1-D grid): workgroups id, id+8, ... share an XCD (private L2).  All query blocks
    // of one (image, head) go to ONE XCD, back to back, so its K / V^T (0.6 MB at T = 2305) is fetched into that
    // L2 once and re-read there by the other query blocks instead of being duplicated in all eight L2s.
    int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (p.dbg & 2) { xcd = 0; idx = blockIdx.x; }
    const int pair = (p.dbg & 2) ? (idx / p.nqb) : (idx / p.nqb) * 8 + xcd;                 // (image, head) pair
    if (pair >= p.B * p.H) return;
    const int qb = idx - (idx / p.nqb) * p.nqb;
    const int b = pair / p.H, h = pair - b * p.H;
    const int q0 = qb * 128 + w * 32;
    const float c = p.scale_log2e;

    // ---- Q fragments (B operand of S^T = K Q^T): lane -> query row q0 + (lane&31), 8 d per chunk
    int qrow = q0 + (lane & 31);
    if (qrow >= p.T) qrow = p.T - 1;
    const bf16_t* qp = p.q + ((int64_t)b * p.Tp + qrow) * p.ld_qk + h * 64;
    bf16x8 qf[4];
#pragma unroll
    for (int kc = 0; kc < 4; kc++) qf[kc] = *(const bf16x8*)(qp + kc * 16 + hi * 8);

    // ---- staging sources: wave w stages rows [w*16, w*16+16) of both tiles (2 DMA each) --------
    const bf16_t* kbase = p.k + (int64_t)b * p.Tp * p.ld_qk + h * 64;
    const bf16_t* vbase = VROW ? p.vt + (int64_t)b * p.Tp * p.ld_qk + h * 64
                               : p.vt + (int64_t)b * p.vt_img_stride + (int64_t)h * 64 * p.Tp;
    // Per-lane byte offsets of this lane's two DMA rows inside a 64-key tile (K) / inside the head's V^T block; the
    // tile's own offset is wave-uniform and is added on the scalar unit, so staging costs no VALU work per tile
    // (the kernel is VALU-bound; the per-tile 64-bit address products were ~10 % of its VALU time).
    // A wave's two pieces of a tile are rows w*8 + (lane>>3) and 32 further: both swizzles have period 16 rows, so the pieces share
    // ONE lane offset per operand and the 32-row step rides in the scalar offset.
    unsigned k_voff, v_voff;
    {
        const int r = w * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        k_voff = (unsigned)((r * p.ld_qk + ch * 8) * 2);
        if constexpr (VROW) v_voff = (unsigned)((r * p.ld_qk + ((lane & 7) ^ swz_vrow(r)) * 8) * 2);    // V row r = key
        else v_voff = (unsigned)((r * p.Tp + ch * 8) * 2);         // V^T row r = d; 8 keys per chunk
    }
    // Full tiles go through `buffer_load_dwordx4 ... offen lds`: the (image, head) base sits in a buffer descriptor, the
    // tile offset in an SGPR and the lane's row/chunk offset in one VGPR computed once -- no VALU work per tile (the
    // global_load_lds form needs a 64-bit VGPR address, i.e. a v_lshl_add_u64 per piece).
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0x7fffffff, 0x00020000);
    const int k_tile_bytes = (int)(64 * p.ld_qk * 2);
    auto stage = [&](int buf, int kv) {                            // full tiles: every key row < T
        unsigned char* base = lds + buf * 16384;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, (int)k_voff, kv * k_tile_bytes + qd * (k_tile_bytes >> 1), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + 8192 + r0 * 128), 16, (int)v_voff,
                                                     VROW ? kv * k_tile_bytes + qd * (k_tile_bytes >> 1) : kv * 128 + qd * 32 * p.Tp * 2, 0, 0);
        }
    };
    auto stage_clamped = [&](int buf, int kv) {                    // the partial last tile: key rows >= T re-read row T-1
        unsigned char* base = lds + buf * 16384;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            const int r = r0 + (lane >> 3);
            const int ch = (lane & 7) ^ ((r >> 1) & 7);
            int key = kv * 64 + r;
            if (key >= p.T) key = p.T - 1;
            // (32-bit buffer offsets: 64-bit per-lane pointers here get hoisted out of the tile loop and spilled across it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, (key * (int)p.ld_qk + ch * 8) * 2, 0, 0, 0);
            if constexpr (VROW)          // same clamp as K: a row past T would be multiplied by P = 0, but must be finite
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + 8192 + r0 * 128), 16,
                                                         (key * (int)p.ld_qk + ((lane & 7) ^ swz_vrow(r)) * 8) * 2, 0, 0, 0);
            else                         // reads past T are finite junk, masked by P = 0
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + 8192 + r0 * 128), 16, (r * p.Tp + ch * 8) * 2, kv * 128, 0, 0);
        }
    };

    // K-only / V-only staging of one tile (pipelined sweep: K runs two tiles ahead of V); `clamp` = the partial last tile
    auto stage_k = [&](int buf, int kv, bool clamp) {
        unsigned char* base = lds + buf * 16384;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            if (!clamp) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, (int)k_voff, kv * k_tile_bytes + qd * (k_tile_bytes >> 1), 0, 0);
            } else {
                const int r = r0 + (lane >> 3);
                const int ch = (lane & 7) ^ ((r >> 1) & 7);
                int key = kv * 64 + r;
                if (key >= p.T) key = p.T - 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LPTR(base + r0 * 128), 16, (key * (int)p.ld_qk + ch * 8) * 2, 0, 0, 0);
            }
        }
    };
    auto stage_v = [&](int buf, int kv, bool clamp) {          // VROW layout only
        unsigned char* base = lds + buf * 16384 + 8192;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = w * 8 + qd * 32;
            if (!clamp) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + r0 * 128), 16, (int)v_voff, kv * k_tile_bytes + qd * (k_tile_bytes >> 1), 0, 0);
            } else {
                const int r = r0 + (lane >> 3);
                int key = kv * 64 + r;
                if (key >= p.T) key = p.T - 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LPTR(base + r0 * 128), 16, (key * (int)p.ld_qk + ((lane & 7) ^ swz_vrow(r)) * 8) * 2, 0, 0, 0);
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[d][r] = 0.f;

    // The SIMD's VALU issue port (~4 cycles per wave64 instruction, shared by its 3 waves; PMC: 78 % busy vs 45 % for
    // the matrix pipe) bounds this kernel, so the softmax is built to need as few VALU instructions as possible:
    //  * Q is scaled by scale*log2(e) ONCE (bf16 round-off of the scaled Q ~ 2^-9 per element, random sign);
    //  * the running maximum is subtracted by the matrix pipe: one extra MFMA per 32x32 score tile whose K-side fragment
    //    is 1.0 in contraction slot 0 and whose Q-side fragment holds -M there (M is kept bf16-exact, so the product is
    //    exact) -- S arrives as s*c - M and P = exp2(S) needs no VALU fma (the matrix pipe is 45 % busy, it has room);
    //  * no per-tile row maximum: the tile keeps the OLD M as long as P cannot overflow -- checked on the tile's row
    //    sums (sum <= 2^40, also catches inf / NaN); only a tile that fails the check takes the
    //    slow path: recompute S with C = 0, explicit maximum, rescale O and l, new splat;
    //  * the offset starts at ZERO (no offset MFMA at all: 16 instead of 18 MFMAs per tile) and is only set by the first tile
    //    that fails the check (overflow, or -- first tile -- a row about to underflow): scores of ordinary size never need one.
    // Per 64-key tile: 32 exp + 32 add + 16 cvt_pk + a handful, instead of ~190 VALU instructions.
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        unsigned wq[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned u = ((const unsigned*)&qf[kc])[e];
            wq[e] = pack_bf2(__uint_as_float(u << 16) * c, __uint_as_float(u & 0xffff0000u) * c);
        }
        const uint4 q4 = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        qf[kc] = __builtin_bit_cast(bf16x8, q4);
    }
    // Offset subtracted from the SCALED scores (log2 domain), bf16-exact.  It starts at 0 -- no offset, and no offset MFMA -- and
    // stays there as long as the row sums neither overflow nor (first tile) underflow: for scores of ordinary size the whole row is
    // exponentiated as it is (f32 / bf16 exponent range), and a tile costs 16 MFMAs instead of 18.  The first tile that fails the
    // check sets it to the row maximum (slow path below); from then on the offset rides in on the extra MFMA as before.
    float M = 0.f;
    bool have_m = false;                                         // wave-uniform: an offset has been set
    // contraction slot 0 (= element 0 of the lanes with hi == 0): K side all ones, Q side -M of the lane's query column
    uint4 ones4 = make_uint4(hi == 0 ? 0x3F80u : 0u, 0u, 0u, 0u);
    bf16x8 kones = __builtin_bit_cast(bf16x8, ones4);
    uint4 qn4 = make_uint4(0u, 0u, 0u, 0u);
    bf16x8 qneg = __builtin_bit_cast(bf16x8, qn4);

    // fragment byte offsets inside a stage buffer, all precomputed once
    // (absolute 32-bit LDS addresses: the stage-buffer offset then folds into the ds_read immediate instead of costing one
    // v_add per read)
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    unsigned k_addr[2][4], v_addr[2][4];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int rk = t * 32 + swap23(lane & 31);
#pragma unroll
        for (int kc = 0; kc < 4; kc++) k_addr[t][kc] = lds0 + rk * 128 + (((kc * 2 + hi) ^ ((rk >> 1) & 7)) << 4);
        const int rv = t * 32 + (lane & 31);
#pragma unroll
        for (int c8 = 0; c8 < 4; c8++) v_addr[t][c8] = lds0 + 8192 + rv * 128 + (((c8 * 2 + hi) ^ ((rv >> 1) & 7)) << 4);   // [d-block t][chunk c8]
        if constexpr (VROW) {
            // transpose-read addresses, [d-block t][half h2 of the lane's 8 keys]: inside its 16-lane group (g = lane>>4) lane j
            // supplies key row (g>>1)*8 + h2*4 + (j>>2), feature quad (j&3) of the group's 16 features t*32 + (g&1)*16 + ..., and
            // receives feature t*32 + (lane&31) for that row's four keys; the 16-key step c8 is an immediate (2 KiB: the swizzle
            // does not depend on it)
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) v_addr[t][h2] = lds0 + 8192 + tr_lane_off(lane, t, h2);
        }
    }
    // opaque to the optimiser, which otherwise re-derives each address from its row and chunk parts at every use
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            asm volatile("" : "+v"(k_addr[t][i]));
            asm volatile("" : "+v"(v_addr[t][i]));
        }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float l_part = 0.f;                                          // this half-wave's running sum of P (unnormalised)

    // one KV tile: S^T = K Q^T (+C), online softmax, O^T += V^T P^T.  BUF is a compile-time buffer index so that the
    // stage offset folds into the ds_read immediate; MASK only for the (peeled) partial last tile; FIRST forces the
    // explicit-maximum path.
    auto tile = [&](auto buf_tag, int kv, auto mask_tag, bool first, auto opt_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr bool NOCHECK = decltype(opt_tag)::value;      // optimistic tile: no offset, no row-sum check, no rescale path
        typedef const __attribute__((address_space(3))) bf16x8* frag_ptr;
        f32x16 s[2];
        // the partial last tile often holds very few keys (T = 2305 = 36*64 + 1): when they all sit in its first 32-key
        // half, the second score tile is skipped altogether (its P is 0)
        const bool half_only = MASK && (p.T - kv * 64 <= 32);
        // the tile's eight K fragments are requested up front: the two score chains then run back to back on counted
        // lgkmcnt waits instead of read -> wait -> MFMA per fragment (and the slow path reuses the registers)
        bf16x8 kfr[2][4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int kc = 0; kc < 4; kc++) kfr[t][kc] = *(frag_ptr)(uintptr_t)(k_addr[t][kc] + BUF * 16384);
        __builtin_amdgcn_sched_barrier(0);
        auto qk = [&](auto sub_tag) {
            constexpr bool sub_max = decltype(sub_tag)::value;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (t == 1 && half_only) continue;
                if (sub_max) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kones, qneg, zero16, 0, 0, 0);    // -M everywhere
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[t][kc], qf[kc], (kc == 0 && !sub_max) ? zero16 : s[t], 0, 0, 0);
                }
            }
            if constexpr (MASK) {
                // lane's key for register r of tile t:  kv*64 + t*32 + 16*(r>>3) + 8*hi + (r&7)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int key = kv * 64 + t * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                        if (key >= p.T || (t == 1 && half_only)) s[t][r] = -INFINITY;
                    }
            }
        };
        bool slow = !NOCHECK && ((p.dbg & 1) || (first && (p.dbg & 4)));    // (bit 2: old behaviour, the first tile always sets the offset)
        float ts = 0.f;
        if constexpr (NOCHECK) {
            qk(std::false_type{});
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    s[t][r] = __builtin_amdgcn_exp2f(s[t][r]);
                    ts += s[t][r];
                }
        } else if (!slow) {
            if (have_m) qk(std::true_type{}); else qk(std::false_type{});      // s = score*c - M   (M = 0: no offset MFMA)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    s[t][r] = __builtin_amdgcn_exp2f(s[t][r]);
                    ts += s[t][r];
                }
            // 2^40: P may have overflowed (or is about to); first tile only: 2^-60, the whole row may be about to underflow
            slow = __any(!(ts <= 1.0995116e12f) || (first && ts < 8.6736174e-19f));
        }
        if (!NOCHECK && slow) {                                  // wave-uniform
            qk(std::false_type{});                               // s = score*c
            float mx = s[0][0];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // bf16-exact (round to nearest: P may exceed 1 by 2^-8, harmless); on the first tile nothing has been accumulated yet
            const float M_new = bf2f(f2bf(first ? mx : fmaxf(M, mx)));
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(M - M_new);
            M = M_new;
            have_m = true;
            l_part *= alpha;
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[d][r] *= alpha;
            qn4.x = hi == 0 ? (unsigned)f2bf(-M) : 0u;
            qneg = __builtin_bit_cast(bf16x8, qn4);
            ts = 0.f;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - M);
                    ts += s[t][r];
                }
        }
        l_part += ts;
        // ---- O^T += V^T P^T (4 chunks of 16 keys, 2 d-blocks) ----
        __builtin_amdgcn_s_setprio(1);                           // favour the wave that feeds the matrix pipe (+1 %)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (t == 1 && half_only) continue;
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const uint4 pw = make_uint4(pack_bf2(s[t][cc * 8 + 0], s[t][cc * 8 + 1]), pack_bf2(s[t][cc * 8 + 2], s[t][cc * 8 + 3]),
                                            pack_bf2(s[t][cc * 8 + 4], s[t][cc * 8 + 5]), pack_bf2(s[t][cc * 8 + 6], s[t][cc * 8 + 7]));
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);     // 4 v_cvt_pk, no repacking
                const int c8 = t * 2 + cc;
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    bf16x8 vf;
                    if constexpr (VROW) {
                        vf = lds_tr8(v_addr[d][0] + c8 * 2048 + BUF * 16384, v_addr[d][1] + c8 * 2048 + BUF * 16384);
                    } else {
                        vf = *(frag_ptr)(uintptr_t)(v_addr[d][c8] + BUF * 16384);
                    }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    const int nkv = (p.T + 63) / 64;
    const int nfull = p.T / 64;          // tiles with no key >= T
    // Every wave must have its LDS reads RETURNED (lgkmcnt(0)), not merely issued, before the barrier: hipcc may
    // sink the MFMAs that consume the tile's last ds_reads below the barrier, and under a loaded LDS pipeline such a
    // read can still be queued when another wave's post-barrier LDS-DMA (250-400 cycles, L2-warm) lands in the same
    // buffer.  Observed as run-to-run differences in ~3 % of rows; tests/test_determinism_gpu.py guards it.
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // a wave whose 32 queries all lie beyond T (the last query block of T = 2305 has ONE valid query: three of its four
    // waves) only takes part in the staging and the barriers
    const bool active = q0 < p.T;
    bool run_classic = true;

    // ---- pipe sweep -----------------------------------------------------------------------------------------------------
    // The classic sweep below runs QK^T -> exp -> PV of ONE tile back to back: inside a wave every phase waits for the one
    // before it, and only the SIMD's other waves fill the gaps (PMC: matrix pipe 54 %, VALU 62 % busy, 39 % of the wave cycles
    // in issue stalls).  Here the three phases of a tile are spread over three iterations so that the matrix work and the VALU
    // work inside one iteration are INDEPENDENT of each other:
    //     iteration i:   matrix pipe:  O += V(i-1)^T P(i-1)^T   and   S(i+1) = K(i+1) Q^T        (16 MFMAs)
    //                    VALU:         P(i) = exp2(S(i)), row sums, bf16 packing                  (~85 instructions)
    // K is staged two tiles ahead of V (K(i+2) and V(i) share stage buffer i&1 while iteration i reads buffer (i+1)&1).
    // The sweep runs WITHOUT a softmax offset (scores of ordinary size: the whole row is exponentiated as it is, exactly the
    // classic sweep's fast path, same operations in the same order -> same bits).  A row sum that overflows (or a first tile
    // about to underflow) only raises a flag here; if any wave of the workgroup raised it, the whole workgroup redoes its query
    // block with the classic sweep, which owns the offset / rescale logic.
    if constexpr (PIPE) {
        static_assert(VROW, "the pipelined sweep reads V row-major");
        typedef const __attribute__((address_space(3))) bf16x8* frag_ptr;
        volatile int* wg_flag = (volatile int*)(lds + 4 * 16384);
        if (threadIdx.x == 0) *wg_flag = 0;
        uint4 pE[4], pO[4];             // packed bf16 probabilities of the even / odd tiles, [16-key chunk]
        bool bad = false;
        const int n = nkv;
        const bool partial = nkv > nfull;
        const bool half_last = partial && (p.T - nfull * 64 <= 32);    // the partial tile's keys all sit in its first half
        auto k_read = [&](auto buf_tag, auto t_tag, bf16x8 (&kf)[4]) {          // the four K fragments of 32-key half t
            constexpr int BUF = decltype(buf_tag)::value, t = decltype(t_tag)::value;
#pragma unroll
            for (int kc = 0; kc < 4; kc++) kf[kc] = *(frag_ptr)(uintptr_t)(k_addr[t][kc] + BUF * 16384);
        };
        auto qk_mma = [&](f32x16& sc, const bf16x8 (&kf)[4]) {
#pragma unroll
            for (int kc = 0; kc < 4; kc++) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kc], qf[kc], kc == 0 ? zero16 : sc, 0, 0, 0);
        };
        // V fragments by hand-issued transpose-reads: behind an LDS-DMA hipcc makes every ds_read_tr builtin wait for vmcnt(0), i.e.
        // for the stage that was issued a few instructions earlier (the classic sweep pays that wait only at the END of a tile).  The
        // asm loads are invisible to that logic; their completion is waited for by `tr_wait` (lgkmcnt(0) with the destination registers
        // as read-write operands, so no consumer can be scheduled above it).  c8 = 16-key chunk, two 32-feature blocks each.
        auto tr_issue = [&](auto buf_tag, auto c8_tag, s16x4_t (&f)[2][2]) {
            constexpr int OFF = decltype(c8_tag)::value * 2048 + decltype(buf_tag)::value * 16384;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const unsigned a0 = v_addr[d][0], a1 = v_addr[d][1];
                s16x4_t lo, hh;
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                             : "=&v"(lo), "=&v"(hh) : "v"(a0), "v"(a1), "i"(OFF));
                f[d][0] = lo; f[d][1] = hh;
            }
        };
        auto tr_wait = [&](s16x4_t (&f)[2][2], s16x4_t (&g)[2][2]) {
            s16x4_t a = f[0][0], b = f[0][1], c = f[1][0], d = f[1][1], e = g[0][0], h = g[0][1], i = g[1][0], j = g[1][1];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(h), "+v"(i), "+v"(j));
            f[0][0] = a; f[0][1] = b; f[1][0] = c; f[1][1] = d; g[0][0] = e; g[0][1] = h; g[1][0] = i; g[1][1] = j;
        };
        auto pv_mma = [&](const uint4& pw, const s16x4_t (&f)[2][2]) {
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const bf16x8 vf = __builtin_shufflevector(f[d][0], f[d][1], 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
            }
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
        // the drain after the last iteration (and nothing else) may skip the second half of a partial tile
        auto pv = [&](auto buf_tag, const uint4 (&pw)[4], bool skip_t1) {
            s16x4_t fa[2][2], fb[2][2];
            tr_issue(buf_tag, C0{}, fa); tr_issue(buf_tag, C1{}, fb);
            tr_wait(fa, fb);
            pv_mma(pw[0], fa); pv_mma(pw[1], fb);
            if (skip_t1) return;
            tr_issue(buf_tag, C2{}, fa); tr_issue(buf_tag, C3{}, fb);
            tr_wait(fa, fb);
            pv_mma(pw[2], fa); pv_mma(pw[3], fb);
        };
        // softmax of one quarter of a tile (8 of the lane's 32 keys): exp2, running row sum (same summation order as the classic
        // sweep: t, then register), bf16 packing -- 8 v_exp + 8 v_add + 4 v_cvt_pk
        auto sm_part = [&](auto t_tag, auto cc_tag, f32x16 (&sc)[2], uint4 (&pw)[4], float& ts) {
            constexpr int t = decltype(t_tag)::value, cc = decltype(cc_tag)::value;
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                e[j] = __builtin_amdgcn_exp2f(sc[t][cc * 8 + j]);
                ts += e[j];
            }
            pw[t * 2 + cc] = make_uint4(pack_bf2(e[0], e[1]), pack_bf2(e[2], e[3]), pack_bf2(e[4], e[5]), pack_bf2(e[6], e[7]));
        };
        // issue-order hint for one segment: four MFMAs, five VALU instructions behind each (a 32x32x16 MFMA occupies the matrix pipe
        // for 32 cycles, about five single-issue instructions of the same wave -- MI355X_MICROARCH.md)
        auto seg_hint = [&]() {
            if constexpr (SCHED) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
        };
        // One iteration = four segments of 4 MFMAs + one softmax quarter (20 VALU) each, hard scheduling boundaries between them (the
        // compiler interleaves inside a segment).  Carried from iteration to iteration: sc[0] = S(i) half 0 (complete), kp = the K
        // fragments of S(i) half 1 whose four MFMAs are still PENDING (they only need registers, so they may run after the barrier that
        // hands the K buffer back to the DMA) -- that way the scores of ONE tile (32 registers) are all that is ever live:
        //   S0  QK(i) half 1 -> sc[1]             | softmax(i) quarter 0 (sc[0])   | V(i-1) reads c0,c1; K(i+1) half-0 reads
        //   S1  PV(i-1) chunks 0,1                | softmax(i) quarter 1 (sc[0])   | V(i-1) reads c2,c3; K(i+1) half-1 reads -> kp
        //   S2  QK(i+1) half 0 -> sc[0] (dead)    | softmax(i) quarter 2 (sc[1])
        //   S3  PV(i-1) chunks 2,3                | softmax(i) quarter 3 (sc[1]), overflow check
        auto body = [&](auto rb_tag, f32x16 (&sc)[2], bf16x8 (&kp)[4], uint4 (&p_out)[4], const uint4 (&p_in)[4], auto pv_tag, auto qk_tag,
                        auto mask_tag, int kv_cur) {
            constexpr bool PV = decltype(pv_tag)::value, QK = decltype(qk_tag)::value, MASK = decltype(mask_tag)::value;
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            auto mask_half = [&](int t) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = kv_cur * 64 + t * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= p.T || (t == 1 && half_last)) sc[t][r] = -INFINITY;
                }
            };
            s16x4_t fa[2][2], fb[2][2];
            bf16x8 k0[4];
            float ts = 0.f;
            // S0
            if constexpr (MASK) mask_half(0);
            if constexpr (PV) { tr_issue(rb_tag, C0{}, fa); tr_issue(rb_tag, C1{}, fb); }
            qk_mma(sc[1], kp);
            if constexpr (QK) k_read(rb_tag, I0{}, k0);
            sm_part(I0{}, I0{}, sc, p_out, ts);
            seg_hint();
            __builtin_amdgcn_sched_barrier(0);
            // S1
            if constexpr (PV) {
                tr_wait(fa, fb);
                pv_mma(p_in[0], fa); pv_mma(p_in[1], fb);
                tr_issue(rb_tag, C2{}, fa); tr_issue(rb_tag, C3{}, fb);
            }
            if constexpr (QK) k_read(rb_tag, I1{}, kp);
            sm_part(I0{}, I1{}, sc, p_out, ts);
            if constexpr (PV) seg_hint();
            __builtin_amdgcn_sched_barrier(0);
            // S2
            if constexpr (MASK) mask_half(1);
            float ts1 = ts;
            sm_part(I1{}, I0{}, sc, p_out, ts1);
            if constexpr (QK) { qk_mma(sc[0], k0); seg_hint(); }
            __builtin_amdgcn_sched_barrier(0);
            // S3
            if constexpr (PV) {
                tr_wait(fa, fb);
                pv_mma(p_in[2], fa); pv_mma(p_in[3], fb);
            }
            sm_part(I1{}, I1{}, sc, p_out, ts1);
            if constexpr (PV) seg_hint();
            // the classic sweep's check: 2^40 (P may have overflowed); first tile only: 2^-60 (the row may be about to underflow)
            bad |= (bool)__any(!(ts1 <= 1.0995116e12f) || (kv_cur == 0 && ts1 < 8.6736174e-19f));
            l_part += ts1;
        };
        using T1 = std::true_type;
        using F0 = std::false_type;
        using B2 = std::integral_constant<int, 2>;
        using B3 = std::integral_constant<int, 3>;
        // Stage ring of FOUR buffers: slot j&3 holds K(j) and V(j-2).  Iteration i reads slot (i+1)&3 and issues K(i+3), V(i+1) into
        // slot (i+3)&3 (last read two iterations ago), so every LDS-DMA piece has two iterations to land and the end-of-iteration wait
        // is a COUNTED vmcnt(4): this iteration's four pieces stay in flight across the barrier.
        auto stage_for = [&](int i) -> int {     // returns the number of pieces issued by this wave
            int np = 0;
            if (i + 3 < n) { stage_k((i + 3) & 3, i + 3, i + 3 >= nfull); np += 2; }
            if (i + 1 < n) { stage_v((i + 3) & 3, i + 1, i + 1 >= nfull); np += 2; }
            return np;
        };
        auto sync_keep4 = [&]() {                // steady state: everything but the newest four pieces has landed
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // dispatch on the (runtime) iteration index to the compile-time (slot, parity) instance: slot read = (i+1)&3, odd i writes pO
        auto run_body = [&](int i, f32x16 (&sc_)[2], bf16x8 (&kp_)[4], auto pv_tag, auto qk_tag, auto mask_tag) {
            switch ((i + 1) & 3) {
                case 0: body(B0{}, sc_, kp_, pO, pE, pv_tag, qk_tag, mask_tag, i); break;       // i = 3 (mod 4)
                case 1: body(B1{}, sc_, kp_, pE, pO, pv_tag, qk_tag, mask_tag, i); break;       // i = 0
                case 2: body(B2{}, sc_, kp_, pO, pE, pv_tag, qk_tag, mask_tag, i); break;       // i = 1
                default: body(B3{}, sc_, kp_, pE, pO, pv_tag, qk_tag, mask_tag, i); break;      // i = 2
            }
        };
        f32x16 sc[2];
        bf16x8 kp[4];
        // prologue: K(0) -> slot 0, K(1) -> slot 1, K(2) + V(0) -> slot 2; S(0) half 0, half-1 fragments pending
        stage_k(0, 0, nfull < 1);
        if (n > 1) stage_k(1, 1, nfull < 2);
        if (n > 2) stage_k(2, 2, nfull < 3);
        stage_v(2, 0, nfull < 1);
        sync();
        {
            bf16x8 k0[4];
            k_read(B0{}, std::integral_constant<int, 0>{}, k0);
            k_read(B0{}, std::integral_constant<int, 1>{}, kp);
            if (active) qk_mma(sc[0], k0);
        }
        // iteration 0: no PV yet
        stage_for(0);
        if (active) {
            if (n == 1) {
                if (partial) run_body(0, sc, kp, F0{}, F0{}, T1{});
                else run_body(0, sc, kp, F0{}, F0{}, F0{});
            } else {
                run_body(0, sc, kp, F0{}, T1{}, F0{});
            }
        }
        sync();
        int i = 1;
        // steady state, four iterations per trip (i = 1 mod 4): every tile touched is a full one -> branch-free staging, compile-time
        // slots, counted waits
        for (; i + 6 < nfull; i += 4) {
            stage_k(0, i + 3, false); stage_v(0, i + 1, false);
            if (active) body(B2{}, sc, kp, pO, pE, T1{}, T1{}, F0{}, i);
            sync_keep4();
            stage_k(1, i + 4, false); stage_v(1, i + 2, false);
            if (active) body(B3{}, sc, kp, pE, pO, T1{}, T1{}, F0{}, i + 1);
            sync_keep4();
            stage_k(2, i + 5, false); stage_v(2, i + 3, false);
            if (active) body(B0{}, sc, kp, pO, pE, T1{}, T1{}, F0{}, i + 2);
            sync_keep4();
            stage_k(3, i + 6, false); stage_v(3, i + 4, false);
            if (active) body(B1{}, sc, kp, pE, pO, T1{}, T1{}, F0{}, i + 3);
            sync_keep4();
        }
        for (; i + 1 < n; i++) {                 // the last few iterations before the final one: clamped staging, plain waits
            stage_for(i);
            if (active) run_body(i, sc, kp, T1{}, T1{}, F0{});
            sync();
        }
        if (n >= 2) {                            // final iteration (i == n-1): PV(n-2), softmax(n-1) (masked if partial), no QK
            if (active) {
                if (partial) run_body(i, sc, kp, T1{}, F0{}, T1{});
                else run_body(i, sc, kp, T1{}, F0{}, F0{});
            }
            sync();
        }
        if (active) {                            // drain: PV(n-1), V(n-1) sits in slot (n+1)&3
            switch ((n + 1) & 3) {
                case 0: pv(B0{}, ((n - 1) & 1) ? pO : pE, half_last); break;
                case 1: pv(B1{}, ((n - 1) & 1) ? pO : pE, half_last); break;
                case 2: pv(B2{}, ((n - 1) & 1) ? pO : pE, half_last); break;
                default: pv(B3{}, ((n - 1) & 1) ? pO : pE, half_last); break;
            }
        }
        // workgroup-uniform verdict (one LDS word behind the stage buffers)
        if (bad && lane == 0) *wg_flag = 1;
        sync();
        run_classic = __builtin_amdgcn_readfirstlane(*wg_flag) != 0;
        if (run_classic) {                       // redo the whole query block with the offset-capable sweep
            sync();
            l_part = 0.f;
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[d][r] = 0.f;
        }
    }

    // ---- optimistic sweep (classic structure, 3 waves per SIMD) -------------------------------------------------------------
    if constexpr (OPT && !PIPE) {
        volatile int* wg_flag = (volatile int*)(lds + 2 * 16384);
        if (threadIdx.x == 0) *wg_flag = 0;
        using T1 = std::true_type;
        using F0 = std::false_type;
        const bool partial = nkv > nfull;
        if (nfull > 0) stage(0, 0); else stage_clamped(0, 0);
        sync();
        // tile 0 through the checked code (first-tile rule: a row about to underflow, or to overflow, sets an offset -> verdict "bad")
        if (nkv > 1) { if (nfull > 1) stage(1, 1); else stage_clamped(1, 1); }
        if (active) { if (nkv == 1 && partial) tile(B0{}, 0, T1{}, true, F0{}); else tile(B0{}, 0, F0{}, true, F0{}); }
        const bool bad0 = have_m;
        sync();
        int kv = 1;
        for (; kv + 1 < nfull; kv += 2) {          // kv odd: buffers 1, 0
            stage(0, kv + 1);
            if (active) tile(B1{}, kv, F0{}, false, T1{});
            sync();
            if (kv + 2 < nfull) stage(1, kv + 2); else if (kv + 2 < nkv) stage_clamped(1, kv + 2);
            if (active) tile(B0{}, kv + 1, F0{}, false, T1{});
            sync();
        }
        if (kv < nfull) {                           // one more full tile (kv odd -> buffer 1)
            if (kv + 1 < nkv) stage_clamped(0, kv + 1);
            if (active) tile(B1{}, kv, F0{}, false, T1{});
            sync();
            kv++;
        }
        if (kv < nkv && active) {                   // the partial tile
            if (kv & 1) tile(B1{}, kv, T1{}, false, T1{}); else tile(B0{}, kv, T1{}, false, T1{});
        }
        // verdict on the accumulated row sums: every tile's row sum is positive, so l_part <= 2^40 implies that no tile's check would
        // have fired (stricter than the per-tile rule: a redo costs time, never bits); inf / NaN fail the comparison too
        const bool bad = bad0 || (active && __any(!(l_part <= 1.0995116e12f)));
        if (bad && lane == 0) *wg_flag = 1;
        sync();
        run_classic = __builtin_amdgcn_readfirstlane(*wg_flag) != 0;
        if (run_classic) {
            sync();
            l_part = 0.f;
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[d][r] = 0.f;
        }
    }

    if (run_classic) {
        if constexpr (PIPE || OPT) {      // (re-)establish the classic sweep's state here, so that none of it is live across the pipelined sweep
            M = 0.f; have_m = false;
            qn4 = make_uint4(0u, 0u, 0u, 0u); qneg = __builtin_bit_cast(bf16x8, qn4);
            ones4 = make_uint4(hi == 0 ? 0x3F80u : 0u, 0u, 0u, 0u); kones = __builtin_bit_cast(bf16x8, ones4);
        }
        if (nfull > 0) stage(0, 0); else stage_clamped(0, 0);
        sync();
        int kv = 0;
        for (; kv + 1 < nfull; kv += 2) {          // two tiles per trip: buffer index is a compile-time constant
            stage(1, kv + 1);
            if (active) tile(B0{}, kv, std::false_type{}, kv == 0, std::false_type{});
            sync();
            if (kv + 2 < nfull) stage(0, kv + 2); else if (kv + 2 < nkv) stage_clamped(0, kv + 2);
            if (active) tile(B1{}, kv + 1, std::false_type{}, false, std::false_type{});
            sync();
        }
        // remainder: at most one full tile and/or the partial tile, buffers alternate from (kv & 1)
        if (kv < nfull) {                           // kv even here -> buffer 0
            if (kv + 1 < nkv) stage_clamped(1, kv + 1);          // kv + 1 == nfull: the partial tile
            if (active) tile(B0{}, kv, std::false_type{}, kv == 0, std::false_type{});
            sync();
            kv++;
        }
        if (kv < nkv && active) {
            if (kv & 1) tile(B1{}, kv, std::true_type{}, false, std::false_type{}); else tile(B0{}, kv, std::true_type{}, kv == 0, std::false_type{});
        }
    }
    const float l_run = l_part;

    // ---- epilogue: normalise, write O (row-major, head h) and LSE ---------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // The epilogue's addresses are formed HERE, from scalars made opaque after the loop: left to itself the compiler computes the
    // per-lane output pointers in the prologue and, at 168 registers, spills them across the whole loop -- 4 VGPRs x 29 184 waves =
    // 30 MB of scratch written and read back per launch (PMC: WRITE_SIZE 143 MB for 114 MB of output).
    int q0e = q0, be = b, he = h;
    asm volatile("" : "+s"(q0e), "+s"(be), "+s"(he));
    const int qr = q0e + (lane & 31);
    uint4 st[2][2];
    pack_token_rows(o, inv, st);                    // 16-byte stores (common.h)
    if (qr < p.T) {
        bf16_t* op = p.out + ((int64_t)be * p.Tp + qr) * p.ld_out + he * 64;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int pr = 0; pr < 2; pr++) *(uint4*)(op + d * 32 + 16 * pr + 8 * hi) = st[d][pr];
        if (p.lse && hi == 0) p.lse[((int64_t)be * p.H + he) * p.Tp + qr] = M + __builtin_amdgcn_logf(l_tot);
    }
}

#ifdef OWL_TUNING   // tuning / race-hunting switches exist only in an OWL_TUNING build (include/owl_hip_tuning.h)
static int g_attn_dbg = 0;
extern "C" int owl_attention_debug(int flags) { g_attn_dbg = flags; return 0; }
#else
static constexpr int g_attn_dbg = 0;
#endif

static constexpr int ATTN_DEFAULT_VARIANT = 1;      // what variant 0 resolves to for row-major V (1 classic, 2 pipelined, 3 pipelined + issue-order hints)

static int attn_fwd_launch(void* stream, const void* q, const void* k, int64_t ld_qk, const void* v, int v_row_major,
                           int64_t vt_img_stride, void* out, int64_t ld_out, float* lse, int64_t B, int64_t H, int64_t T,
                           int64_t Tp, float scale, int variant) {
    OWL_CHECK_ARG(q && k && v && out, "owl_attention_fwd_bf16: null pointer");
    OWL_CHECK_ARG(ld_qk % 8 == 0 && ld_out % 8 == 0 && Tp % 8 == 0 && T > 0 && T <= Tp, "owl_attention_fwd_bf16: bad strides (ld_qk %% 8, ld_out %% 8, Tp %% 8)");
    AttnFwdP p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.ld_qk = ld_qk;
    p.vt = (const bf16_t*)v; p.vt_img_stride = vt_img_stride;
    p.out = (bf16_t*)out; p.ld_out = ld_out; p.lse = lse;
    p.T = (int)T; p.Tp = (int)Tp; p.H = (int)H;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.B = (int)B; p.nqb = (int)((T + 127) / 128); p.dbg = g_attn_dbg;
    const int64_t npairs8 = (B * H + 7) / 8;                  // pairs per XCD (rounded up)
    dim3 grid((unsigned)(npairs8 * p.nqb * 8));
    if (variant == 4 && v_row_major) {
        hipLaunchKernelGGL((attn_fwd_kernel<true, false, false, true>), grid, dim3(256), 2 * 16384 + 16, (hipStream_t)stream, p);
        OWL_LAUNCH_CHECK();
        return 0;
    }
    OWL_CHECK_ARG(variant >= 0 && variant <= 3 && (variant <= 1 || v_row_major), "owl_attention_fwd: variant must be 0 (default), 1 (classic), 2 / 3 (pipelined; row-major V only)");
    if (variant == 0) variant = ATTN_DEFAULT_VARIANT;
    if (!v_row_major) hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, dim3(256), 2 * 16384, (hipStream_t)stream, p);
    else if (variant == 2 || variant == 3) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384 + 16);
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384 + 16);
            attr_done = true;
        }
        if (variant == 2) hipLaunchKernelGGL((attn_fwd_kernel<true, true, false>), grid, dim3(256), 4 * 16384 + 16, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<true, true, true>), grid, dim3(256), 4 * 16384 + 16, (hipStream_t)stream, p);
    }
    else hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, dim3(256), 2 * 16384, (hipStream_t)stream, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

extern "C" int owl_attention_fwd_bf16(void* stream, const void* q, const void* k, int64_t ld_qk, const void* vt,
                                      int64_t vt_img_stride, void* out, int64_t ld_out, float* lse, int64_t B,
                                      int64_t H, int64_t T, int64_t Tp, float scale) {
    return attn_fwd_launch(stream, q, k, ld_qk, vt, 0, vt_img_stride, out, ld_out, lse, B, H, T, Tp, scale, 0);
}

// same, with V read where the QKV GEMM leaves it: row-major [B*Tp, ld_qkv], head h at column h*64 of `v` (no V^T copy at all)
extern "C" int owl_attention_fwd_vrow_bf16(void* stream, const void* q, const void* k, const void* v, int64_t ld_qkv, void* out,
                                           int64_t ld_out, float* lse, int64_t B, int64_t H, int64_t T, int64_t Tp, float scale, int variant) {
    return attn_fwd_launch(stream, q, k, ld_qkv, v, 1, 0, out, ld_out, lse, B, H, T, Tp, scale, variant);
}
