// Fused encoder self-attention forward, ONE WAVE PER SIMD: 64 query rows per wave, 256 per workgroup (one workgroup per CU).
// Reference arithmetic: HF5:377-402 eager_attention_forward -- softmax(Q K^T * dh^-0.5) V, no mask, dh = 64.
//
// Why a second structure (cdna_hip_programming.md "4-wave, one-wave-per-SIMD, persistent structure"): the 3-waves-per-SIMD
// kernel (attention_fwd.hip) spends one LDS fragment KB per MFMA for 32 queries per wave and leaves the matrix pipe 57 % busy
// (profiles/r03_pmc.md).  Here a wave owns TWO 32-query blocks that share every K and V fragment (half the LDS fragment reads and half the
// LDS-DMA pieces per query), and the QK^T MFMAs of tile j+1 / the PV MFMAs of tile j are interleaved with the softmax of tiles j, j+1 inside
// the ONE instruction stream of the SIMD's only wave:
//     phase A(j): 16 MFMAs S(j+1)^T = K(j+1) Q^T   ||  exp2 / row sums / bf16 packing of the second half of S(j)   ||  16 V(j) transpose-reads
//     phase B(j): 16 MFMAs O^T += V(j)^T P(j)^T    ||  exp2 / row sums / bf16 packing of the first half of S(j+1)  ||  8 K(j+2) fragment reads
// One barrier per tile; K two tiles / V one tile ahead by LDS-DMA into 2-deep rings (issued from inline asm so that hipcc neither drains them
// in front of the transpose-reads nor counts them -- waits are hand-placed).
//
// Softmax: the same exp2-domain form as the classic kernel's FAST path (Q pre-scaled by scale*log2 e once, offset 0, no per-tile maximum), but
// with ONE verdict per query block instead of a check per tile: a block whose row sums left [2^-60, 2^60] (or are inf / NaN), or whose class-token
// score is outside +-40, raises its entry of `redo`; the launcher then runs the classic kernel over exactly those blocks (its explicit-maximum
// slow path).  Scores of ordinary size never take it.
//
// Tiling: class token peeled exactly like attn_fwd_kernel<true, true> (token 0 = initial state of every other query's softmax and one VALU-only
// workgroup per (image, head)); tiles cover tokens 1..T-1, which must be a multiple of 64 keys (B/16: 2304 = 36 tiles = 9 query blocks).
#ifdef OWL_TUNING   // whole file: a tuning-build experiment, not part of the shipped library
#include "attention_fwd_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// LDS-DMA piece (1 KiB per wave-instruction) from inline asm: M0 = wave-uniform LDS byte address, written in the statement that uses it.
__device__ __forceinline__ void dma_piece(const i32x4_t& rsrc, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// The hot loop's MFMAs are inline asm with explicit register files: hipcc (ROCm 7.2) puts EVERY MFMA result of a > 256-register kernel into
// the accumulator file, and a score that lives there costs one v_accvgpr_read per exp2 -- 64 more VALU issues per tile in a loop that is bound
// by exactly those.  So: scores S in arch VGPRs (the softmax reads them directly), O in AGPRs (touched by MFMAs only; an "a" constraint on the
// Q fragments makes hipcc copy them from VGPRs in front of every MFMA instead of keeping them there, so they stay VGPR operands).
// hipcc neither schedules nor pads an asm statement: the order below is the issue order (one sched_barrier per MFMA slot), and every MFMA
// result is consumed at least a phase (16 MFMAs) later -- except the accumulate chains, which need no wait states.
__device__ __forceinline__ void mfma_s_first(f32x16& s, const bf16x8& kf, const bf16x8& qf) {      // S = K Q^T (C = 0)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf), "v"(qf));
}
__device__ __forceinline__ void mfma_s_acc(f32x16& s, const bf16x8& kf, const bf16x8& qf) {        // S += K Q^T
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "v"(qf));
}
__device__ __forceinline__ void mfma_o_acc(f32x16& o, const bf16x8& vf, const bf16x8& pf) {        // O^T += V^T P^T
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pf));
}
#define SLOT_END() __builtin_amdgcn_sched_barrier(0)
// hipcc does not know that the asm statements above are MFMAs: a register copy it places behind one (it moves O and S between the registers
// two instantiations of the tile step chose -- at the loop exit, between the tail steps) would read an accumulator the matrix pipe has not
// written back yet (8-pass MFMA -> any reader but the next MFMA of its chain: 12 wait states).  Wherever control leaves the steady-state
// loop body this pad comes first; inside the loop the step is one instantiation pair with O and S in fixed registers (audited in the ISA:
// no v_accvgpr_* / v_mov of an MFMA result between the loop's barriers; tests hold the output bitwise to the classic kernel's).
#define MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")

struct AttnW64P {
    AttnFwdP c;          // q / k / vt (= row-major V) / out / lse / T / Tp / H / B / scale_log2e as the classic kernel; nqb = 256-query blocks + 1
    int* redo;           // [pairs][nqb - 1]: 1 = this query block must be redone by the classic kernel (written for EVERY block)
};

template <int DBG>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(AttnW64P pp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const AttnFwdP& p = pp.c;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    // XCD-aware mapping as the classic kernel: all query blocks of one (image, head) on ONE XCD, back to back (its K / V stay in that L2)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int pair = (idx / p.nqb) * 8 + xcd;
    if (pair >= p.B * p.H) return;
    const int qb = idx - (idx / p.nqb) * p.nqb;
    const int b = pair / p.H, h = pair - b * p.H;
    if (qb == p.nqb - 1) { attn_cls_row(p, b, h, lds); return; }       // workgroup-uniform
    const int Tk = p.T - 1;                        // tiles cover tokens 1 .. T-1
    const int q0 = qb * 256 + w * 64;
    const bool active = q0 < Tk;
    const float c = p.scale_log2e;
    const int64_t ld = p.ld_qk;

    // ---- Q fragments of the wave's two 32-query blocks (B operand of S^T = K Q^T), pre-scaled -----------------------------------------
    bf16x8 qf[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        int qrow = q0 + blk * 32 + (lane & 31);
        if (qrow >= Tk) qrow = Tk - 1;
        const bf16_t* qp = p.q + ((int64_t)b * p.Tp + qrow + 1) * ld + h * 64;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
            const uint4 u4 = *(const uint4*)(qp + kc * 16 + hi * 8);
            const unsigned u[4] = {u4.x, u4.y, u4.z, u4.w};
            unsigned wq[4];
#pragma unroll
            for (int e = 0; e < 4; e++) wq[e] = pack_bf2(__uint_as_float(u[e] << 16) * c, __uint_as_float(u[e] & 0xffff0000u) * c);
            qf[blk][kc] = __builtin_bit_cast(bf16x8, make_uint4(wq[0], wq[1], wq[2], wq[3]));
        }
    }
    // row 0 of K and V (the class token as a key)
    uint4 k0u[4];
    unsigned short v0u[2];
    {
        const bf16_t* k0p = p.k + (int64_t)b * p.Tp * ld + h * 64;
        const bf16_t* v0p = p.vt + (int64_t)b * p.Tp * ld + h * 64;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) k0u[kc] = *(const uint4*)(k0p + kc * 16 + hi * 8);
#pragma unroll
        for (int d = 0; d < 2; d++) v0u[d] = v0p[d * 32 + (lane & 31)];
    }

    // ---- staging: wave w brings rows [w*8, w*8+8) and 32 further of every K / V tile (2 + 2 pieces), 16-byte chunks swizzled as the classic kernel
    const bf16_t* kbase = p.k + ((int64_t)b * p.Tp + 1) * ld + h * 64;
    const bf16_t* vbase = p.vt + ((int64_t)b * p.Tp + 1) * ld + h * 64;
    auto make_rsrc = [](const void* ptr) {
        const uint64_t a = (uint64_t)(uintptr_t)ptr;
        i32x4_t r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));      // stride 0, no swizzle
        r.z = 0x7fffffff;
        r.w = 0x00020000;
        return r;
    };
    const i32x4_t k_rsrc = make_rsrc(kbase), v_rsrc = make_rsrc(vbase);
    unsigned k_voff, v_voff;
    {
        const int r = w * 8 + (lane >> 3);
        k_voff = (unsigned)((r * ld + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
        v_voff = (unsigned)((r * ld + ((lane & 7) ^ swz_vrow(r)) * 8) * 2);
    }
    const int tile_bytes = (int)(64 * ld * 2);
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    constexpr int KST = 8192, VOFF = 2 * 8192;              // K stages at 0 / 8192, V stages at 16384 / 24576
    const unsigned dst_row = __builtin_amdgcn_readfirstlane(lds0 + w * 8 * 128);
    auto stage_k = [&](int st, int kv) {
#pragma unroll
        for (int qd = 0; qd < 2; qd++) dma_piece(k_rsrc, dst_row + st * KST + qd * 4096, k_voff, kv * tile_bytes + qd * (tile_bytes >> 1));
    };
    auto stage_v = [&](int st, int kv) {
#pragma unroll
        for (int qd = 0; qd < 2; qd++) dma_piece(v_rsrc, dst_row + VOFF + st * KST + qd * 4096, v_voff, kv * tile_bytes + qd * (tile_bytes >> 1));
    };

    // ---- fragment addresses (absolute LDS byte addresses; stage / chunk offsets are immediates) ----------------------------------------
    unsigned k_addr[2][4], v_addr[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int rk = t * 32 + swap23(lane & 31);
#pragma unroll
        for (int kc = 0; kc < 4; kc++) k_addr[t][kc] = lds0 + rk * 128 + (((kc * 2 + hi) ^ ((rk >> 1) & 7)) << 4);
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) v_addr[t][h2] = lds0 + VOFF + tr_lane_off(lane, t, h2);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(k_addr[t][i]));
#pragma unroll
        for (int i = 0; i < 2; i++) asm volatile("" : "+v"(v_addr[t][i]));
    }
    typedef const __attribute__((address_space(3))) bf16x8* frag_ptr;

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[2][2];                     // [d block][query block]: O^T accumulators
    float l[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // [query block][chain]: this half-wave's running sums of P
    bool bad = false;                   // wave-uniform: the classic kernel must redo this block

    // all waves: barriers with every LDS read RETURNED and every own DMA piece landed (a 2-deep ring: the pieces were issued a tile ago)
    auto sync = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int n = Tk >> 6;              // key tiles (>= 3: launcher)
    bf16x8 kfr[2][4];                   // K fragments of the tile whose scores are computed next
    bf16x8 vfr[2][4];                   // V^T fragments of the tile whose PV runs next: [d block][16-key chunk]
    unsigned pw[2][4][4];               // P^T fragments (bf16 pairs) of the current tile: [query block][16-key chunk][word]
    f32x16 sa[2][2], sb[2][2];          // scores / probabilities [query block][32-key block] of two tiles in flight

    auto read_k = [&](int st, int i) {      // i = 0..7 -> (t, kc)
        kfr[i >> 2][i & 3] = *(frag_ptr)(uintptr_t)(k_addr[i >> 2][i & 3] + st * KST);
    };
    // softmax piece k = 0..15 of the half `t` of s: two exp2, two row-sum adds, one packed conversion
    auto soft_piece = [&](f32x16 (&s)[2][2], int t, int k) {
        const int blk = k >> 3, r = (k & 7) * 2;
        const float e0 = __builtin_amdgcn_exp2f(s[blk][t][r]), e1 = __builtin_amdgcn_exp2f(s[blk][t][r + 1]);
        l[blk][0] += e0;
        l[blk][1] += e1;
        pw[blk][t * 2 + (r >> 3)][(r & 7) >> 1] = pack_bf2(e0, e1);
    };
    // the same with the row-sum contribution scaled by a wave-uniform 1.0 / 0.0 (fma with 1.0 = the add, bit for bit): the last tile step runs
    // the uniform loop body for a tile that does not exist, whose "probabilities" are never used and must not be summed
    auto soft_piece_m = [&](f32x16 (&s)[2][2], int t, int k, float m) {
        const int blk = k >> 3, r = (k & 7) * 2;
        const float e0 = __builtin_amdgcn_exp2f(s[blk][t][r]), e1 = __builtin_amdgcn_exp2f(s[blk][t][r + 1]);
        l[blk][0] = __builtin_fmaf(e0, m, l[blk][0]);
        l[blk][1] = __builtin_fmaf(e1, m, l[blk][1]);
        pw[blk][t * 2 + (r >> 3)][(r & 7) >> 1] = pack_bf2(e0, e1);
    };

    // ---- prologue -------------------------------------------------------------------------------------------------------------------------
    stage_k(0, 0); stage_k(1, 1); stage_v(0, 0);
    // key 0 as the initial state, on the matrix pipe (as attn_fwd_kernel<true, true>): s0 by four MFMAs against a K fragment whose only non-zero
    // row is k0; O = p0 v0 by MFMAs with contraction slot 0 the only live one
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const bool row0 = (lane & 31) == 0;
        f32x16 s0t = zero16;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
            const uint4 kz = row0 ? k0u[kc] : make_uint4(0u, 0u, 0u, 0u);
            s0t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kz), qf[blk][kc], s0t, 0, 0, 0);
        }
        const float s0 = __shfl(s0t[0], lane & 31, 64);
        if (__any(!(fabsf(s0) <= 40.f))) bad = true;
        const float p0 = __builtin_amdgcn_exp2f(s0);
        l[blk][0] = hi == 0 ? p0 : 0.f;
        const uint4 pz = make_uint4(hi == 0 ? (unsigned)f2bf(p0) : 0u, 0u, 0u, 0u);
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const uint4 vz = make_uint4(hi == 0 ? (unsigned)v0u[d] : 0u, 0u, 0u, 0u);
            o[d][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vz), __builtin_bit_cast(bf16x8, pz), zero16, 0, 0, 0);
        }
    }
    sync();
    // S(0) -> sa, first half of its softmax; K(1) fragments; K(2) -> stage 0 once everybody has read K(0)
#pragma unroll
    for (int i = 0; i < 8; i++) read_k(0, i);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int kc = i >> 2, t = (i >> 1) & 1, blk = i & 1;
        if (kc == 0) mfma_s_first(sa[blk][t], kfr[t][kc], qf[blk][kc]); else mfma_s_acc(sa[blk][t], kfr[t][kc], qf[blk][kc]);
    }
    MFMA_DRAIN();
    sync();                              // everybody holds its K(0) fragments: stage 0 is free for K(2)
    stage_k(0, 2);
#pragma unroll
    for (int i = 0; i < 8; i++) read_k(1, i);
#pragma unroll
    for (int k = 0; k < 16; k++) soft_piece(sa, 0, k);

    // ---- one tile step ------------------------------------------------------------------------------------------------------------------------
    // cur = S(j) (first half already exponentiated and packed), nxt receives S(j+1).  ONE body for every tile (two instantiations: ring stage 0 / 1,
    // both inside the one loop below): hipcc keeps O, S and the fragments in fixed registers around a loop, but between separately inlined copies
    // of the step (peeled tail steps were the first version) it re-shuffles them with v_mov / v_accvgpr copies placed right behind the asm MFMAs
    // -- which it does not know to be MFMAs: no wait states, stale accumulators (seen as NaN / wrong query blocks).  So the last step runs the
    // same body: the MFMAs of the tile that does not exist are skipped by a wave-uniform branch per slot, the row sums of that tile's first half
    // are multiplied by 0 (its registers hold stale scores), its probabilities are never used, its K fragments come from a stale ring stage;
    // the K / V staging is guarded at run time.  (Letting those MFMAs run on the stale fragments instead -- no branches -- gave reproducibly
    // wrong row sums in the wave's FIRST query block; not understood, measured 4 % faster, not shipped.)
    auto step = [&](f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto st_tag, int j) {
        constexpr int st = decltype(st_tag)::value;          // = j & 1: ring stage of K(j+2) and V(j)
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const bool stamp = DBG == 8 && blockIdx.x == 0 && j >= 16 && j < 20;
        if (DBG == 8 && stamp) t0 = __builtin_amdgcn_s_memtime();
        sync();
        if (DBG == 8 && stamp) t1 = __builtin_amdgcn_s_memtime();
        if (j + 3 < n) stage_k(st ^ 1, j + 3);
        if (j + 1 < n) stage_v(st ^ 1, j + 1);
        if (!active) return;
        if (DBG == 8 && stamp) t2 = __builtin_amdgcn_s_memtime();
        const float m = (j + 1 < n) ? 1.f : 0.f;
        // ---- phase A: QK^T of tile j+1  ||  second half of softmax(j)  ||  V(j) transpose-reads.  One MFMA slot = the MFMA + 2 exp2 + 2 adds +
        //      1 packed conversion + 1 transpose-read (6 single-issue fillers per 32-cycle MFMA)
        s16x4_t vlo[2][4], vhh[2][4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            {
                const int kc = i >> 2, t = (i >> 1) & 1, blk = i & 1;      // four independent accumulate chains
                if (j + 1 < n) { if (kc == 0) mfma_s_first(nxt[blk][t], kfr[t][kc], qf[blk][kc]); else mfma_s_acc(nxt[blk][t], kfr[t][kc], qf[blk][kc]); }
            }
            soft_piece(cur, 1, i);
            {
                const int c8 = i >> 2, d = (i >> 1) & 1, h2 = i & 1;
                const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)(v_addr[d][h2] + c8 * 2048 + st * KST));
                if (h2 == 0) vlo[d][c8] = v; else vhh[d][c8] = v;
            }
            SLOT_END();
        }
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int c8 = 0; c8 < 4; c8++) vfr[d][c8] = __builtin_shufflevector(vlo[d][c8], vhh[d][c8], 0, 1, 2, 3, 4, 5, 6, 7);
        if (DBG == 8 && stamp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t3 = __builtin_amdgcn_s_memtime(); }
        // ---- phase B: PV of tile j  ||  first half of softmax(j+1)  ||  K(j+2) fragment reads
        bf16x8 pf[2][4];
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int c8 = 0; c8 < 4; c8++) pf[blk][c8] = __builtin_bit_cast(bf16x8, make_uint4(pw[blk][c8][0], pw[blk][c8][1], pw[blk][c8][2], pw[blk][c8][3]));
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c8 = i >> 2, d = (i >> 1) & 1, blk = i & 1;
            mfma_o_acc(o[d][blk], vfr[d][c8], pf[blk][c8]);
            if (DBG == 8 && stamp && j == 17 && w == 0 && lane == 0) ((unsigned*)pp.redo)[8192 + 64 + i] = (unsigned)(__builtin_amdgcn_s_memtime() - t3);
            soft_piece_m(nxt, 0, i, m);
            if ((i & 1) == 0) read_k(st, i >> 1);
            SLOT_END();
        }
        if (DBG == 8 && stamp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t4 = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                unsigned* tr = (unsigned*)pp.redo + 8192 + ((j - 16) * 4 + w) * 4;
                tr[0] = (unsigned)(t1 - t0); tr[1] = (unsigned)(t2 - t1); tr[2] = (unsigned)(t3 - t2); tr[3] = (unsigned)(t4 - t3);
            }
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int j = 0; j < n; j += 2) {         // even tiles: scores in sa, ring stage 0; odd tiles: sb, stage 1
        step(sa, sb, S0{}, j);
        if (j + 1 < n) step(sb, sa, S1{}, j + 1);
    }
    MFMA_DRAIN();

    // ---- verdict + epilogue -----------------------------------------------------------------------------------------------------------------
    float lt[2];
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const float lr = l[blk][0] + l[blk][1];
        lt[blk] = lr + __shfl_xor(lr, 32, 64);
        if (active && __any(!(lt[blk] <= 1.1529215e18f) || lt[blk] < 8.6736174e-19f)) bad = true;
    }
    // one flag per query block: every block writes it (0 or 1), so the caller never has to clear the array
    {
        int* flag_lds = (int*)lds;
        sync();                                     // (all fragment reads done: the LDS is free)
        if (threadIdx.x == 0) *flag_lds = 0;
        __syncthreads();
        if (bad && lane == 0) atomicOr(flag_lds, 1);
        __syncthreads();
        if (threadIdx.x == 0) pp.redo[pair * (p.nqb - 1) + qb] = *flag_lds;
    }
    if (!active) return;
    int q0e = q0, be = b, he = h;
    asm volatile("" : "+s"(q0e), "+s"(be), "+s"(he));
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const int qr = q0e + blk * 32 + (lane & 31) + 1;        // token
        const float inv = 1.0f / lt[blk];
        const f32x16 ob[2] = {o[0][blk], o[1][blk]};
        uint4 st4[2][2];
        pack_token_rows(ob, inv, st4);
        if (qr < p.T) {
            bf16_t* op = p.out + ((int64_t)be * p.Tp + qr) * p.ld_out + he * 64;
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int pr = 0; pr < 2; pr++) *(uint4*)(op + d * 32 + 16 * pr + 8 * hi) = st4[d][pr];
            if (p.lse && hi == 0) p.lse[((int64_t)be * p.H + he) * p.Tp + qr] = __builtin_amdgcn_logf(lt[blk]);
        }
    }
}

// host side: attention_fwd.hip (attn_fwd_w64_launch is called from its launcher)
int attn_fwd_w64_launch(hipStream_t stream, const AttnFwdP& base, int* redo, int dbg) {
    AttnW64P pp;
    pp.c = base;
    pp.c.nqb = (base.T - 1 + 255) / 256 + 1;
    pp.redo = redo;
    const int64_t npairs8 = ((int64_t)base.B * base.H + 7) / 8;
    const dim3 grid((unsigned)(npairs8 * pp.c.nqb * 8));
#ifdef OWL_TUNING
    if (dbg == 8) hipLaunchKernelGGL(attn_fwd_w64_kernel<8>, grid, dim3(256), 4 * 8192, stream, pp);        // s_memtime stamps (tools/attn_w64_trace.py)
    else
#endif
    hipLaunchKernelGGL(attn_fwd_w64_kernel<0>, grid, dim3(256), 4 * 8192, stream, pp);
    OWL_LAUNCH_CHECK();
    return 0;
}
#endif  // OWL_TUNING (whole file)
