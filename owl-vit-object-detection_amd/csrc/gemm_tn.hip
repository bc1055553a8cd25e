// TN split-K GEMM for the weight gradients:  slab[s][n][k] = sum_{m in split s} dY[m][n] * X[m][k]
// (dW = dY^T X, ref: autograd of every nn.Linear on the trainable path; HF5:437-439,457,472,474,994-997; ref models.py:25).
//
// Both operands are read the way the forward/backward kernels leave them in HBM -- token-major [M, features] -- so the
// contraction index m is the SLOW index of both tiles.  The first version of this repo made token-contiguous copies
// (16 transpose launches, 3.6 GB of HBM traffic, 1.3 ms per step) and used the NT kernel; here the transposition is done
// by the LDS hardware instead: `ds_read_b64_tr_b16` gives a lane 4 consecutive m of ONE feature column (probed on the
// device, tools/probe/tr_probe.hip: inside a 16-lane group lane j supplies the address of row j>>2 / 8-byte column
// quad j&3 of a [4 m][16 feature] block and receives column j), two of them = the 8-element K-fragment of
// v_mfma_f32_32x32x16_bf16.
//
// LDS image of a K-tile (64 m) per operand: [64 rows][256 features] bf16, 512-byte rows, staged by LDS-DMA (one 1-KiB
// instruction = 2 rows).  A transpose-read touches rows {k..k+3, k+8..k+11} x 64 B, which at a 512-byte pitch would all
// sit on the same 16 banks; the 16-byte chunks of row r are therefore XOR-swizzled with ((r&3)<<2) on the DMA source
// side (and in the read addresses), which spreads the 4 rows of a block over the 4 quarters of the 256-byte bank row:
// 32 pieces of 16 B on 16 slots = the 2-cycle minimum of a 512-byte access.
//
// Tile 256 (n) x 256 (k) x 64 (m), 8 waves (2 x 4, wave tile 128 x 64), 2 x 64 KiB LDS, one barrier per K-tile,
// persistent (tile, split) items, f32 slabs through the LDS-staged coalesced epilogue of gemm_common.h.
// Rows m >= M of the last K-tile are fetched from a caller-provided zero row, so any M works.
#include "gemm_common.h"

static constexpr int TBK = 64;
static constexpr int T_OP_BYTES = TBK * 256 * 2;          // 32 KiB per operand per stage
static constexpr int T_STAGE = 2 * T_OP_BYTES;
static constexpr int T_LDS = 2 * T_STAGE;

struct TnP {
    const bf16_t* dY; int64_t ldy;     // [M, ldy], features n in [0, N)
    const bf16_t* X; int64_t ldx;      // [M, ldx], features k in [0, K)
    const bf16_t* zero_row;            // >= 512 bytes of zeros
    float* slab; int64_t slab_stride;  // [nsplit][N][K]
    float* bias_slab;                  // optional [nsplit][N]: column sums of dY over the split's token range (the bias gradient's partial sums), see gemm_tn_pp_kernel
    int64_t M, N, K;
    int tiles_n, tiles_k, kt_per_split, nsplit, persistent;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// 8-element K-fragment = two transpose-reads (the compiler's builtin: it tracks lgkmcnt itself and builds the 128-bit
// register tuple without copies -- an asm ds_read's result may be touched before a hand-placed wait)
__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned char* tile_lo, const unsigned char* tile_hi) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)tile_lo);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)tile_hi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(512) void gemm_tn_slab_kernel(TnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int nk_all = (int)((p.M + TBK - 1) / TBK);
    const int nitems = p.tiles_n * p.tiles_k * p.nsplit;
    int item, item_end, item_step;
    if (p.persistent) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, chunk = (nitems + 7) >> 3;
        item = xcd * chunk + idx; item_end = min(nitems, (xcd + 1) * chunk); item_step = gridDim.x >> 3;
    } else {
        item = xcd_remap(blockIdx.x, nitems); item_end = item + 1; item_step = 1;
    }
    if (item >= item_end) return;

    int64_t n0, k0;
    int split, kt0, kt1;
    // items are split-major: consecutive items (= the workgroups of one XCD at any time) are different output tiles of
    // the SAME token range, so the dY / X rows they stream are shared through that XCD's L2 (tile-major order read every
    // operand byte ~2x: FETCH_SIZE 1.0 GB per launch at 4.6 TB/s)
    const int ntiles = p.tiles_n * p.tiles_k;
    auto decode = [&](int it) {
        split = it / ntiles;
        const int tile = it - split * ntiles;
        const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
        n0 = (int64_t)tn * 256; k0 = (int64_t)tk * 256;
        kt0 = split * p.kt_per_split;
        kt1 = min(nk_all, kt0 + p.kt_per_split);
    };

    // ---- staging: wave w fills rows [w*8, w*8+8) of both operand tiles: 4 DMA instructions (2 rows each) per operand ----
    // lane -> row (lane>>5) of the pair, 16-byte chunk (lane&31); the SOURCE chunk is swizzled with ((row&3)<<2)
    auto stage = [&](int buf, int kt) {
        unsigned char* base = lds + buf * T_STAGE;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = w * 8 + q * 2 + (lane >> 5);
            const int c = (lane & 31) ^ ((r & 3) << 2);
            const int64_t m = (int64_t)kt * TBK + r;
            const bool ok = m < p.M;
            int64_t nn = n0 + c * 8; if (nn + 8 > p.N) nn = p.N - 8;
            int64_t kk = k0 + c * 8; if (kk + 8 > p.K) kk = p.K - 8;
            const bf16_t* ga = ok ? p.dY + m * p.ldy + nn : p.zero_row + (lane & 31) * 8;
            const bf16_t* gb = ok ? p.X + m * p.ldx + kk : p.zero_row + (lane & 31) * 8;
            __builtin_amdgcn_global_load_lds(GPTR(ga), LPTR(base + (w * 8 + q * 2) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GPTR(gb), LPTR(base + T_OP_BYTES + (w * 8 + q * 2) * 512), 16, 0, 0);
        }
    };

    // ---- transpose-read addresses (bytes inside an operand tile, K-chunk kc = 0, first half): the lane supplies
    //      row 8*(lane>>5) + ((lane&15)>>2) (+16*kc + 4*half as an immediate), feature f = f0 + 16*((lane>>4)&1) + 4*(lane&3)
    const int rr = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int sw = (rr & 3) << 2;                               // = ((lane&15)>>2) << 2: unchanged by +16*kc, +4*half
    const unsigned lds_base = 0;
    unsigned a_addr[4], b_addr[2];                             // byte offsets from `lds`
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int f = wr * 128 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        a_addr[i] = lds_base + rr * 512 + ((((f >> 3) ^ sw)) << 4) + ((f >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int f = wc * 64 + j * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        b_addr[j] = lds_base + T_OP_BYTES + rr * 512 + ((((f >> 3) ^ sw)) << 4) + ((f >> 2) & 1) * 8;
    }

    decode(item);
    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;
    GemmP ep{};                       // what the shared f32 epilogue needs
    ep.out = p.slab; ep.ldo = p.K; ep.M = p.N; ep.N = p.K; ep.alpha = 1.0f; ep.slab_stride = p.slab_stride;
    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        const int64_t cn0 = n0, ck0 = k0;
        const int csplit = split, ckt0 = kt0, ckt1 = kt1;
        const int next = item + item_step;
        const bool has_next = next < item_end;
        for (int kt = ckt0; kt < ckt1; kt++) {
            const bool last = (kt + 1 == ckt1);
            if (!last) stage(cur ^ 1, kt + 1);
            else if (has_next) { decode(next); stage(cur ^ 1, kt0); }
            const unsigned char* tb = lds + cur * T_STAGE;
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                bf16x8 fa[4], fb[2];
#pragma unroll
                for (int j = 0; j < 2; j++) fb[j] = lds_tr_frag(tb + b_addr[j] + kc * 8192, tb + b_addr[j] + kc * 8192 + 2048);
#pragma unroll
                for (int i = 0; i < 4; i++) fa[i] = lds_tr_frag(tb + a_addr[i] + kc * 8192, tb + a_addr[i] + kc * 8192 + 2048);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D rows = k, cols = n
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        {
            // buffer cur^1 was just consumed; each wave stages its f32 passes through the 2 x 4 KiB of it that only IT will
            // DMA into next (rows [w*8, w*8+8) of both operand tiles)
            unsigned char* xb = lds + (cur ^ 1) * T_STAGE;
            unsigned char* pieceA = xb + (w * 8) * 512;
            unsigned char* pieceB = xb + T_OP_BYTES + (w * 8) * 512;
#pragma unroll
            for (int i = 0; i < 4; i++)
                epi_pass<EPI_SLAB_F32>(ep, acc[i][0], acc[i][1], pieceA, pieceB, cn0 + wr * 128 + i * 32, ck0 + wc * 64, lane, csplit);
        }
        if (!has_next) break;
        item = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Ping-pong schedule for the same tile (round 2): the K-tile body of gemm_pp.hip -- two wave groups (waves 0-3 / 4-7 = output rows n 0-127 /
// 128-255, one wave of each per SIMD) one barrier apart, four quadrant phases per K-tile each split into a LOAD half (fragment reads + LDS-DMA
// requests, lgkmcnt(0) before the barrier) and an MFMA half (8 MFMAs), counted vmcnt -- with the fragments coming out of the token-major
// tiles by transpose-reads.  A token row of the LDS image serves every output column, so the X tile (the "B" side, read in q0 / q1) is free
// once both groups have run their q1 LOAD and the dY tile (read in q0 / q2) once both have run q2: K-tile c+2 is requested into the buffer K-tile c
// is being computed from (X pieces in q2, dY pieces in q3, then the counted wait that retires K-tile c+1) -- more than a K-tile of flight.
// hipcc puts a vmcnt(0) in front of every ds_read_tr BUILTIN that follows an LDS-DMA, which would drain the requests just made in every LOAD
// half: the transpose-reads are hand-issued (asm) and waited for with one lgkmcnt(0) tied to their destination registers.
// One (tile, split) item per workgroup (the host sizes the splits for ~256 items), so no cross-item stream; the f32 epilogue is the shared
// LDS-staged one, in the stage buffers after the last K-tile.  Same accumulation order as the kernel above: identical bits.
// ---------------------------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void tnp_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void tnp_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// the K-fragment (8 tokens of one feature column) = two transpose-reads; `lo` / `hi` land in adjacent registers of the MFMA operand
#define TNP_TR2(lo, hi, addr, OFF) \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF), "n"((OFF) + 2048))

__global__ __launch_bounds__(512) void gemm_tn_pp_kernel(TnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = w >> 2, wc = w & 3;                       // group = 128-row half of the output tile (rows = dY features n)
    const int nk_all = (int)((p.M + TBK - 1) / TBK);
    const int nitems = p.tiles_n * p.tiles_k * p.nsplit;
    const int item = xcd_remap(blockIdx.x, nitems);
    if (item >= nitems) return;
    const int ntiles = p.tiles_n * p.tiles_k;
    const int split = item / ntiles;
    const int tile = item - split * ntiles;
    const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
    const int64_t n0 = (int64_t)tn * 256, k0 = (int64_t)tk * 256;
    const int kt0 = split * p.kt_per_split, kt1 = min(nk_all, kt0 + p.kt_per_split);
    const int nkt = kt1 - kt0;                                 // >= 1
    // Round 6: the bias gradient rides along.  db[n] = sum_m dY[m][n] used to be a kernel of its own (colsum_bf16: dY read a second time out of the HBM, 38 ... 83 us
    // per Linear, and +1.0 % on the train step when all four are skipped -- profiles/r06_colsum_ablate.log).  The dY^T fragments of the MFMAs already hold exactly
    // these values: lane l of a fragment owns feature n = 32 i + (l & 31) and 8 consecutive tokens, so ONE wave per row group (wc == 0) of the workgroups of the
    // first k-tile column (tk == 0) adds its fragments up with v_dot2c_f32_bf16 against (1, 1) -- 64 instructions per K-tile in the shadow of 32 MFMAs -- and
    // writes one partial per (split, feature); owl_slab_reduce adds the splits in fixed order like the weight slabs.
    const bool do_cs = p.bias_slab != nullptr && tk == 0 && wc == 0;      // (wave-uniform)
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- staging (as above): wave w fills token rows [w*8, w*8+8) of both operand tiles, 4 pieces (2 rows each) per operand --------------
    int s_kt = kt0, s_buf = 0;                                 // next K-tile to request, and its buffer
    // lane offsets inside a K-tile, computed once (the K-tile's own offset is wave-uniform: no address VALU per piece); only a K-tile that
    // reaches past row M -- the last one of the whole token range -- takes the per-lane path that substitutes the zero row
    unsigned y_voff[4], x_voff[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = w * 8 + q * 2 + (lane >> 5);
        const int c = (lane & 31) ^ ((r & 3) << 2);
        int64_t nn = n0 + c * 8; if (nn + 8 > p.N) nn = p.N - 8;
        int64_t kk = k0 + c * 8; if (kk + 8 > p.K) kk = p.K - 8;
        y_voff[q] = (unsigned)((r * p.ldy + nn) * 2);
        x_voff[q] = (unsigned)((r * p.ldx + kk) * 2);
    }
    auto stage_one = [&](bool x_side) {                        // 4 VMEM ops
        unsigned char* base = lds + s_buf * T_STAGE + (x_side ? T_OP_BYTES : 0);
        if ((int64_t)(s_kt + 1) * TBK <= p.M) {
            const unsigned char* g = x_side ? (const unsigned char*)(p.X + (int64_t)s_kt * TBK * p.ldx) : (const unsigned char*)(p.dY + (int64_t)s_kt * TBK * p.ldy);
#pragma unroll
            for (int q = 0; q < 4; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + (x_side ? x_voff[q] : y_voff[q])), LPTR(base + (w * 8 + q * 2) * 512), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = w * 8 + q * 2 + (lane >> 5);
            const int c = (lane & 31) ^ ((r & 3) << 2);
            const int64_t m = (int64_t)s_kt * TBK + r;
            const bool ok = m < p.M;
            const bf16_t* g;
            if (x_side) { int64_t kk = k0 + c * 8; if (kk + 8 > p.K) kk = p.K - 8; g = ok ? p.X + m * p.ldx + kk : p.zero_row + (lane & 31) * 8; }
            else { int64_t nn = n0 + c * 8; if (nn + 8 > p.N) nn = p.N - 8; g = ok ? p.dY + m * p.ldy + nn : p.zero_row + (lane & 31) * 8; }
            __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(base + (w * 8 + q * 2) * 512), 16, 0, 0);
        }
    };
    auto stage_advance = [&]() { s_buf ^= 1; s_kt++; };
    auto stream_live = [&]() { return s_kt < kt1; };

    // ---- transpose-read addresses (absolute LDS addresses of buffer 0; the buffer and the K-chunk ride in immediates / one add) ------------
    const int rr = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int sw = (rr & 3) << 2;
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    unsigned a_addr[4], b_addr[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int f = grp * 128 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        a_addr[i] = lds0 + rr * 512 + ((((f >> 3) ^ sw)) << 4) + ((f >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int f = wc * 64 + j * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        b_addr[j] = lds0 + T_OP_BYTES + rr * 512 + ((((f >> 3) ^ sw)) << 4) + ((f >> 2) & 1) * 8;
    }

    // prologue: K-tiles 0 and 1 requested, K-tile 0 landed
    stage_one(true); stage_one(false); stage_advance();
    if (stream_live()) { stage_one(true); stage_one(false); stage_advance(); tnp_wait<8>(); } else { tnp_wait<0>(); }
    tnp_bar();

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    int cur = 0;
    if (grp == 1) tnp_bar();                                   // group 1 runs one barrier behind
    for (int kt = 0; kt < nkt; kt++) {
        const unsigned bo = (unsigned)(cur * T_STAGE);
        bf16x8 fa[2][4], fb[2][4];                              // [tile of the quadrant][kc], [j][kc]
        // A fragment = two transpose-reads into the two halves of one 128-bit MFMA operand.  The asm returns the halves as separate 64-bit
        // values; they are joined ONCE, right behind the wait that covers them (joined at every use the register allocator copies each
        // fragment into a fresh tuple: 144 v_mov per K-tile in the first version).
        auto ld4 = [&](bf16x8 (&f)[4], unsigned ad) {
            s16x4_t l0, h0, l1, h1, l2, h2, l3, h3;
            TNP_TR2(l0, h0, ad, 0); TNP_TR2(l1, h1, ad, 8192); TNP_TR2(l2, h2, ad, 16384); TNP_TR2(l3, h3, ad, 24576);
            f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7); f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            f[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7); f[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto ld_a = [&](int ih) { ld4(fa[0], a_addr[2 * ih] + bo); ld4(fa[1], a_addr[2 * ih + 1] + bo); };
        auto ld_b = [&](int j) { ld4(fb[j], b_addr[j] + bo); };
        // the reads of a LOAD half have returned: lgkmcnt(0) tied to every fragment register of the K-tile (no consumer can move above it)
        auto wait_frags = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]),
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]));
        };
        auto mma = [&](int ih, int j) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kc = 0; kc < 4; kc++)
#pragma unroll
                for (int t = 0; t < 2; t++)
                    acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kc], fa[t][kc], acc[2 * ih + t][j], 0, 0, 0);   // D rows = k, cols = n
            __builtin_amdgcn_s_setprio(0);
        };
        auto colsum = [&](int ih) {                              // fa[t][kc]: 4 dwords = 8 tokens of the lane's feature
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
            typedef __attribute__((ext_vector_type(2))) short s2_t;
            const bf2_t ones = __builtin_bit_cast(bf2_t, 0x3f803f80u);
            // (the four token pairs of a fragment by CONSTANT shuffles of the fragment itself: subscripting a bit-cast copy of the asm-produced tuple made hipcc 7.2
            //  read its first dword four times -- caught by tools/debug_bias_fold.py with dY[m][n] = m / 64)
#define TN_CS_PAIR(F, A, B) __builtin_bit_cast(bf2_t, (s2_t)__builtin_shufflevector(F, F, A, B))
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    float c = cs[2 * ih + t];
                    c = __builtin_amdgcn_fdot2_f32_bf16(TN_CS_PAIR(fa[t][kc], 0, 1), ones, c, false);
                    c = __builtin_amdgcn_fdot2_f32_bf16(TN_CS_PAIR(fa[t][kc], 2, 3), ones, c, false);
                    c = __builtin_amdgcn_fdot2_f32_bf16(TN_CS_PAIR(fa[t][kc], 4, 5), ones, c, false);
                    c = __builtin_amdgcn_fdot2_f32_bf16(TN_CS_PAIR(fa[t][kc], 6, 7), ones, c, false);
                    cs[2 * ih + t] = c;
                }
#undef TN_CS_PAIR
        };
        // ---- q0 ----
        ld_b(0); ld_a(0);
        wait_frags(); tnp_bar();
        mma(0, 0);
        if (do_cs) colsum(0);
        tnp_bar();
        // ---- q1 ----
        ld_b(1);
        wait_frags(); tnp_bar();
        mma(0, 1);
        tnp_bar();
        // ---- q2: the X tile of this buffer is free (both groups have read both column fragments) ----
        ld_a(1);
        const bool live = stream_live();
        if (live) stage_one(true);
        wait_frags(); tnp_bar();
        mma(1, 1);
        if (do_cs) colsum(1);
        tnp_bar();
        // ---- q3: the dY tile is free; retire K-tile kt+1, leave kt+2 in flight ----
        if (live) { stage_one(false); stage_advance(); tnp_wait<8>(); } else { tnp_wait<0>(); }
        tnp_bar();
        mma(1, 0);
        tnp_bar();
        cur ^= 1;
    }
    if (do_cs) {                                               // token halves (lanes l, l + 32) -> one partial per feature and split
#pragma unroll
        for (int i = 0; i < 4; i++) {
            cs[i] += __shfl_xor(cs[i], 32, 64);
            const int64_t n = n0 + grp * 128 + i * 32 + (lane & 31);
            if (lane < 32 && n < p.N) p.bias_slab[(int64_t)split * p.N + n] = cs[i];
        }
    }
    if (grp == 0) tnp_bar();                                   // group 1's last MFMA half
    {
        // every request has landed and every fragment has been read (the loop ends in waits + barriers): the stage buffers are free, each
        // wave stages its f32 passes through 2 x 4 KiB of its own
        GemmP ep{};
        ep.out = p.slab; ep.ldo = p.K; ep.M = p.N; ep.N = p.K; ep.alpha = 1.0f; ep.slab_stride = p.slab_stride;
        unsigned char* pieceA = lds + (w * 8) * 512;
        unsigned char* pieceB = lds + T_OP_BYTES + (w * 8) * 512;
#pragma unroll
        for (int i = 0; i < 4; i++)
            epi_pass<EPI_SLAB_F32>(ep, acc[i][0], acc[i][1], pieceA, pieceB, n0 + grp * 128 + i * 32, k0 + wc * 64, lane, split);
    }
}

OWL_API int owl_gemm_tn_slab_workspace_bytes(int64_t M, int64_t N, int64_t K, int splits, int64_t* bytes) {
    OWL_CHECK_ARG(bytes && M > 0 && N > 0 && K > 0 && splits >= 1, "owl_gemm_tn_slab_workspace_bytes: bad arguments");
    const int nk = (int)((M + TBK - 1) / TBK);
    if (splits > nk) splits = nk;
    const int per = (nk + splits - 1) / splits;
    *bytes = (int64_t)((nk + per - 1) / per) * N * K * (int64_t)sizeof(float);
    return 0;
}

OWL_API int owl_gemm_tn_slab_bf16(void* stream, const void* dY, int64_t ldy, const void* X, int64_t ldx, const void* zero_row,
                                     float* slab, int64_t M, int64_t N, int64_t K, int splits, int* splits_used, int variant, float* bias_slab) {
    OWL_CHECK_ARG(dY && X && zero_row && slab && splits_used, "owl_gemm_tn_slab_bf16: null pointer");
    OWL_CHECK_ARG(!bias_slab || variant != 1, "owl_gemm_tn_slab_bf16: bias_slab (column sums of dY from the same pass) exists in the ping-pong kernel only (variant 0 / 2)");
    OWL_CHECK_ARG(M > 0 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0, "owl_gemm_tn_slab_bf16: bad M=%lld N=%lld K=%lld (N, K %% 8 == 0)", (long long)M, (long long)N, (long long)K);
    OWL_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && splits >= 1, "owl_gemm_tn_slab_bf16: ldy/ldx must be multiples of 8, splits >= 1");
    TnP p{};
    p.dY = (const bf16_t*)dY; p.ldy = ldy; p.X = (const bf16_t*)X; p.ldx = ldx; p.zero_row = (const bf16_t*)zero_row;
    p.slab = slab; p.slab_stride = N * K; p.M = M; p.N = N; p.K = K; p.bias_slab = bias_slab;
    p.tiles_n = (int)((N + 255) / 256); p.tiles_k = (int)((K + 255) / 256);
    const int nk = (int)((M + TBK - 1) / TBK);
    if (splits > nk) splits = nk;
    p.kt_per_split = (nk + splits - 1) / splits;
    p.nsplit = (nk + p.kt_per_split - 1) / p.kt_per_split;
    *splits_used = p.nsplit;
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_tn_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    });
    OWL_CHECK_ARG(variant == 0 || variant == 1 || variant == 2, "owl_gemm_tn_slab_bf16: variant must be 0 (automatic), 1 (single-phase) or 2 (ping-pong)");
    const int nitems = p.tiles_n * p.tiles_k * p.nsplit;
    if (variant != 1) {
        static unsigned long long attr2_done = 0;
        OWL_ONCE_PER_DEVICE(attr2_done, {
            (void)hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
        });
        hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3(nitems), dim3(512), T_LDS, (hipStream_t)stream, p);
        OWL_LAUNCH_CHECK();
        return 0;
    }
    p.persistent = nitems > 256 ? 1 : 0;
    hipLaunchKernelGGL(gemm_tn_slab_kernel, dim3(p.persistent ? 256 : nitems), dim3(512), T_LDS, (hipStream_t)stream, p);
    OWL_LAUNCH_CHECK();
    return 0;
}
