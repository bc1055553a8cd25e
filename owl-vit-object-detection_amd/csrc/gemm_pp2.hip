// Two-phase ping-pong variant of the 256x256x64 bf16 GEMM (every epilogue but the transposing one): same tile, same LDS image, same epilogues and the
// same accumulation order as gemm_pp.hip (identical bits) -- a different split of the K-tile into phases.
//
// gemm_pp.hip runs a K-tile as FOUR quadrant phases of 8 MFMAs on TWO accumulator tiles: two dependent chains, which one wave issues at
// ~36 cycles per MFMA instead of 32 (a chain's next MFMA needs the previous one's result: `s_memtime` traces, tools/probe), and eight
// barriers per K-tile.  Here a K-tile is TWO phases of 16 MFMAs on FOUR accumulator tiles (four independent chains, four deep):
//     phase A: rows i0,i1 x columns j0,j1     LOAD A: A(i0,i1) + B(j0) + B(j1) fragments (16 ds_read_b128) + this wave's 4 A pieces
//     phase B: rows i2,i3 x columns j0,j1     LOAD B: A(i2,i3) fragments (8; B stays in registers)           + this wave's 4 B pieces
// The two groups (waves 0-3 / 4-7 = output rows 0-127 / 128-255, one wave of each per SIMD) run one barrier apart, as there.
// LDS lifetime: a buffer's B rows are free once both groups have run LOAD A, its A rows once both have run LOAD B.  So the B pieces run TWO
// K-tiles ahead (requested in LOAD B of K-tile c into the buffer c is being computed from) and the A pieces ONE K-tile ahead (requested in
// LOAD A of K-tile c into the other buffer, whose A rows K-tile c-1 released one barrier earlier): two independent DMA cursors that cross
// tile boundaries of the persistent loop.  One counted wait per K-tile, at the end of LOAD B: everything but the B pieces just requested has
// landed -- that is K-tile c+1 complete, for both groups, before the first barrier after which anybody reads it.
// (A second version gave every group its own A rows to stage -- rows i0,i1 two K-tiles ahead, rows i2,i3 one ahead, two counted waits per
// K-tile, five to six half-slots of flight for every piece instead of two for the A pieces -- and measured 4-5 % slower than this one on the
// model's shapes: the extra cursor and wait cost more than the longer flight buys.  profiles/r02_gemm_two_phase.md)
#include "gemm_common.h"
#include <type_traits>

static constexpr int QBM = 256, QBN = 256, QBK = 64;
static constexpr int Q_A_BYTES = QBM * QBK * 2, Q_B_BYTES = QBN * QBK * 2, Q_STAGE = Q_A_BYTES + Q_B_BYTES;   // 32 + 32 KiB
static constexpr int Q_BIAS_OFF = 2 * Q_STAGE, Q_LDS = 2 * Q_STAGE + 2 * 1024;

template <int N> __device__ __forceinline__ void q_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void q_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void q_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// TRACE (OWL_TUNING builds, tools/pp2_trace.py): workgroup 0 stamps s_memtime at every LOAD / wait / barrier / MFMA boundary of K-tile 4 of its first tile
// LINES: the W tile is staged with its rows permuted inside every 64-row group so that a lane's accumulators are 64 contiguous output bytes, and the
// epilogue stores quad-contiguous (gemm_common.h, epi_lines_bf16): forward bf16 epilogues.
template <int EPI, bool TRACE = false, bool LINES = false>
__global__ __launch_bounds__(512) void gemm_pp2_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int grp = w >> 2, wc = w & 3;
    const int nk = (int)(p.K / QBK);                 // >= 2 (host checks)
    const int nitems = p.tiles_m * p.tiles_n;
    // item -> (row tile, column tile): column tiles in blocks of `bw` (the largest divisor of tiles_n up to 4), row tiles inside a block, the block's
    // columns fastest.  An XCD's 32 workgroups then sit on ~8 row panels x bw column tiles: the block's W tiles (bw x 393 KiB at K = 768) stay in that
    // XCD's 4 MiB L2 for the whole sweep over the row panels, instead of all tiles_n of them (4.7 MiB at N = 3072) being re-fetched every round.
    const int bw = p.nsplit;                         // (the launcher passes the block width here: a GEMM on this kernel has no K splits)
    auto decode = [&](int it, int& tm, int& tn) {
        const int per_block = p.tiles_m * bw;
        const int cb = it / per_block, rem = it - cb * per_block;
        tm = rem / bw; tn = cb * bw + (rem - tm * bw);
    };
    int item, item_end, item_step;
    if (p.persistent) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, chunk = (nitems + 7) >> 3;
        item = xcd * chunk + idx; item_end = min(nitems, (xcd + 1) * chunk); item_step = gridDim.x >> 3;
    } else {
        item = xcd_remap(blockIdx.x, nitems); item_end = item + 1; item_step = 1;
    }
    if (item >= item_end) return;
    // Start stagger (round 6 experiment, p.stagger > 0): every second workgroup of an XCD sleeps p.stagger x ~8 k cycles before its first tile, so that half of the
    // chip's epilogues (HBM store / load bursts of 128 KiB per CU) fall into the other half's K-loops instead of all 256 CUs bursting at once.
    if (p.stagger > 0 && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < p.stagger; i++) __builtin_amdgcn_s_sleep(127);
    // (OWL_TUNING builds, timing only -- results are wrong: p.dbg bit 16 = no LDS-DMA requests after the prologue, bit 17 = fragments read in a tile's first
    //  K-tile only, bit 18 = no epilogue; tools/gemm_pp2_ablate.py, profiles/r04_gemm_fr.md section 5)
#ifdef OWL_TUNING
    const bool abl_nodma = (p.dbg >> 16) & 1, abl_noread = (p.dbg >> 17) & 1, abl_noepi = (p.dbg >> 18) & 1;
    bool abl_prologue = true;
#else
    constexpr bool abl_nodma = false, abl_noread = false, abl_noepi = false, abl_prologue = true;
#endif

    // ---- two DMA cursors over the K-tiles in consumption order (across the persistent tile loop) ----------------------------
    int a_item = item, a_k = 0, a_buf = 0;                     // A pieces: one K-tile ahead
    int b_item = item, b_k = 0, b_buf = 0, b_parity = 0;       // B pieces (+ the tile's bias slice): two K-tiles ahead
    unsigned a_voff[2][2], w_voff[2][2], b_voff = 0;           // [half][q]: wave w stages rows half*128 + (w*2+q)*8 + (lane>>3) of a half-tile
    const bf16_t* a_base = nullptr;
    const bf16_t* w_base = nullptr;
    const float* b_base = nullptr;
    const bool has_bias = p.bias != nullptr;
    // EPI_PATCH_F32: A is gathered straight from the image: row m = (b, py, px), k = (channel, ky, kx); a 16-byte chunk is 8 pixels of one patch row.
    // Power-of-two patch sizes: a K-tile is 64 / ps patch rows of ONE channel -- the lane's part of the address (patch origin + the chunk's row / column
    // inside the K-tile) is K-tile-invariant, the K-tile's part (channel, first row) is wave-uniform: same scheme as a plain A matrix.
    // Other patch sizes (L/14: 14-pixel rows = 28 bytes): the K index pads every patch row to psp = 2^n positions (14 -> 16, K = 3 * 14 * 16 = 672 -> 704) and
    // chunk cc of a row starts at pixel min(8 cc, ps - 8): the last chunk OVERLAPS its predecessor instead of running past the row (positions 8, 9 of a 14-pixel
    // row are pixels 6, 7 again) and the weight matrix holds zeros at the duplicated positions (weights.patch_weight_gather_layout) -- no read outside the
    // patch, no garbage-times-zero.  A K-tile's 64 / psp rows may straddle two channels, so the row part of the address (channel, ky) is formed per K-tile and
    // lane from the row index R = K-tile * rows + lane's row (R / ps by multiply-shift).  The 16-byte pieces are only 4-byte aligned in HBM (28 px + 12 bytes).
    constexpr bool GATHER = (EPI == EPI_PATCH_F32);
    const bool np2 = GATHER && p.ps != (int64_t(1) << p.ps_log2);      // (uniform)
    int a_j[2] = {0, 0};                                        // np2: the lane's row inside a K-tile, per q (the chunk swizzle depends on q, not on the half)
    auto stage_A = [&]() {                                     // 4 VMEM ops
        if (a_k == 0) {
            int tm, tn_unused; decode(a_item, tm, tn_unused);
            const int64_t m0 = (int64_t)tm * QBM;
            a_base = GATHER ? p.A : p.A + m0 * p.lda;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int r = h * 128 + (w * 2 + q) * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    int64_t am = m0 + r; if (am >= p.a_rows) am = p.a_rows - 1;
                    if constexpr (GATHER) {
                        // (32-bit arithmetic: the 64-bit divisions here cost 14 spilled VGPRs inside the K loop; unsigned byte offsets: the host checks that the image batch stays below 2^32 bytes, csrc/gemm.hip)
                        const unsigned P32 = (unsigned)p.P, G32 = (unsigned)p.G, S32 = (unsigned)p.S, ps32 = (unsigned)p.ps;
                        const unsigned b = (unsigned)am / P32, pp = (unsigned)am - b * P32;
                        const unsigned py = pp / G32, px = pp - py * G32;
                        unsigned ky = (unsigned)(c * 8) >> p.ps_log2, kx = (unsigned)(c * 8) & ((1u << p.ps_log2) - 1u);
                        if (np2) { a_j[q] = (int)ky; ky = 0; kx = min(kx, ps32 - 8u); }        // (row part added per K-tile; the last chunk overlaps instead of overrunning)
                        a_voff[h][q] = ((((b * 3u) * S32 + py * ps32 + ky) * S32) + px * ps32 + kx) * 2u;
                    } else {
                        a_voff[h][q] = (unsigned)(((am - m0) * p.lda + c * 8) * 2);
                    }
                }
        }
        unsigned char* base = lds + a_buf * Q_STAGE;
        int64_t koff = (int64_t)a_k * QBK;                     // elements from a_base to the K-tile
        unsigned roff[2] = {0u, 0u};                           // np2: byte offset of the lane's patch row (channel, ky) of this K-tile, per q
        if constexpr (GATHER) {
            if (np2) {
                koff = 0;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    unsigned R = (unsigned)(a_k * (QBK >> p.ps_log2) + a_j[q]);
                    R = min(R, 3u * (unsigned)p.ps - 1u);                     // (the K padding's rows: any row of the patch, their weights are zero)
                    const unsigned ch = (R * (unsigned)p.ps_magic) >> 16, ky = R - ch * (unsigned)p.ps;
                    roff[q] = ((ch * (unsigned)p.S + ky) * (unsigned)p.S) * 2u;
                }
            } else {
                const int k = a_k * QBK;
                const int ch = k >> (2 * p.ps_log2), ky = (k & ((1 << (2 * p.ps_log2)) - 1)) >> p.ps_log2;
                koff = ((int64_t)ch * p.S + ky) * p.S;
            }
        }
        const unsigned char* g = (const unsigned char*)(a_base + koff);
        if (!abl_nodma || abl_prologue) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + (a_voff[h][q] + roff[q])), LPTR(base + (h * 128 + (w * 2 + q) * 8) * 128), 16, 0, 0);
        }
        a_buf ^= 1;
        if (++a_k == nk) { a_k = 0; a_item += item_step; }
    };
    auto stage_B = [&]() -> int {                              // 4 or 5 VMEM ops
        const bool first = b_k == 0;
        if (first) {
            int tm, tn; decode(b_item, tm, tn);
            const int64_t n0 = (int64_t)tn * QBN;
            w_base = p.W + n0 * p.ldw;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int r = h * 128 + (w * 2 + q) * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    // LINES: LDS row (j, qd, hi, e) of a 64-row group <- W row 32 hi + 16 j + 4 qd + e of the group (the XOR key stays the LDS row's)
                    const int rs = LINES ? ((r & ~63) | (((r >> 2) & 1) << 5) | (((r >> 5) & 1) << 4) | (((r >> 3) & 3) << 2) | (r & 3)) : r;
                    int64_t wn = n0 + rs; if (wn >= p.w_rows) wn = p.w_rows - 1;
                    w_voff[h][q] = (unsigned)(((wn - n0) * p.ldw + c * 8) * 2);
                }
            if (has_bias) {
                int64_t n = n0 + lane * 4; if (n + 4 > p.N) n = p.N - 4;
                b_base = p.bias + n0;
                b_voff = (unsigned)((n - n0) * 4);
            }
        }
        unsigned char* base = lds + b_buf * Q_STAGE + Q_A_BYTES;
        const unsigned char* g = (const unsigned char*)(w_base + (int64_t)b_k * QBK);
        int n_ops = 0;
        if (!abl_nodma || abl_prologue) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + w_voff[h][q]), LPTR(base + (h * 128 + (w * 2 + q) * 8) * 128), 16, 0, 0);
        n_ops = 4;
        if (first && has_bias) {
            __builtin_amdgcn_global_load_lds(GPTR((const unsigned char*)b_base + b_voff), LPTR(lds + Q_BIAS_OFF + b_parity * 1024), 16, 0, 0);
            n_ops = 5;
        }
        }
        b_buf ^= 1;
        if (++b_k == nk) { b_k = 0; b_item += item_step; b_parity ^= 1; }
        return n_ops;
    };

    const int a_row0 = grp * 128 + (lane & 31), b_row0 = wc * 64 + (lane & 31);
    const int a_base_off = a_row0 * 128, b_base_off = Q_A_BYTES + b_row0 * 128;
    const int a_swz = (a_row0 >> 1) & 7, b_swz = (b_row0 >> 1) & 7;

    // prologue: B(0) (+ bias), A(0), B(1) requested; K-tile 0 landed (B(1) may stay in flight)
    stage_B(); stage_A(); stage_B();
    q_wait<4>();
    q_bar();
#ifdef OWL_TUNING
    abl_prologue = false;
#endif

    int cur = 0, tile_parity = 0;
    const bool tr_wg = TRACE && blockIdx.x == 0;
    bool tr_first = false;          // armed at the workgroup's (p.dbg >> 8)-th tile: the first tile of a launch runs at boost clock with cold queues
    int tr_tile = 0;
    if (TRACE && (p.dbg >> 8) == 0) tr_first = true;
    unsigned long long tr_ts[12] = {};
    bool a_early = false;            // the A pieces of the next tile's K-tile 1 went out before this tile's epilogue stores
    int pending_stores = 0;          // epilogue stores issued after them (known counts only; otherwise the epilogue is followed by a full wait)
    // Store-bound epilogue (bias): group 1 runs ONE barrier interval behind group 0 for the WHOLE persistent loop -- the epilogue is just
    // another interval: group 0 converts and stores its half of a tile while group 1 runs its last MFMA phase, group 1 stores while group 0
    // already reads the next tile's first fragments (+2 % on top of the two-phase K-tile at K = 768 / 3072).  VALU-bound epilogue (quick-GELU:
    // an exp and a reciprocal per element): two staggered epilogues in a row cost more than both together (-6 %), so there the groups are
    // re-synchronised at every tile as in gemm_pp.hip -- group 0 waits for group 1's last MFMA phase, both run their epilogues in one interval.
    // (With quad-contiguous stores -- LINES -- the bias epilogue is VALU-bound too and goes with the second kind: same box, QKV 243.6 -> 233.5 us,
    //  out-proj 89.1 -> 87.2, step 1217 -> 1223 img/s.)
    constexpr bool STAGGERED_EPI = ((EPI == EPI_BIAS_BF16 && !LINES) || EPI == EPI_F32 || EPI == EPI_ACC_F32 || EPI == EPI_PATCH_F32 || EPI == EPI_PATCHM_F32);
    if (STAGGERED_EPI && grp == 1) q_bar();
    unsigned long long wg_t0 = 0;
    unsigned long long wg_r0 = 0;          // s_memrealtime: 100 MHz, one counter for the chip -- comparable across workgroups
    if constexpr (TRACE) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0), "=s"(wg_r0) :: "memory");
    unsigned long long tile_ts[5] = {};
    auto tile_stamp = [&](int i) {
        if constexpr (TRACE) { if (tr_wg && tr_first) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tile_ts[i]) :: "memory"); }
    };
    while (true) {
        tile_stamp(0);
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        int tm, tn; decode(item, tm, tn);
        const int64_t cm0 = (int64_t)tm * QBM, cn0 = (int64_t)tn * QBN;
        if (!STAGGERED_EPI && grp == 1) q_bar();     // (re-)create the one-barrier offset
        tile_stamp(1);
        for (int kt = 0; kt < nk; kt++) {
            const unsigned char* tb = lds + cur * Q_STAGE;
            auto stamp = [&](int idx) {               // (issued without waiting for the result: a waited stamp costs ~140 cycles)
                if constexpr (TRACE) {
                    if (tr_wg && tr_first && kt == ((p.dbg >> 4) & 15)) {
                        // (stamps 0 / 2 / 7 / 10 sit where no LDS read is in flight and are waited for: the un-waited result of a stamp in front of the
                        //  staging code is lost when hipcc moves its SGPR pair before the value has landed; 1 and 6 are not recorded)
                        if (idx == 1 || idx == 6) return;
                        if (idx == 0 || idx == 2 || idx == 7 || idx >= 10) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_ts[idx]));
                        else asm volatile("s_memtime %0" : "=s"(tr_ts[idx]));
                    }
                }
            };
            bf16x8 fa[2][4], fb[2][4];                // [row tile of the phase][kc], [j][kc]
            auto ld_a = [&](int ih) {
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int kc = 0; kc < 4; kc++)
                        fa[t][kc] = *(const bf16x8*)(tb + a_base_off + (2 * ih + t) * 4096 + (((kc * 2 + hi) ^ a_swz) << 4));
            };
            auto mma = [&](int ih) {                   // 16 MFMAs, four independent chains (kc outer: every chain sees kc = 0..3 in order)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kc], fa[t][kc], acc[2 * ih + t][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };
            // ---- phase A ----
            stamp(0);
            if (!abl_noread || kt == 0) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int kc = 0; kc < 4; kc++) fb[j][kc] = *(const bf16x8*)(tb + b_base_off + j * 4096 + (((kc * 2 + hi) ^ b_swz) << 4));
            ld_a(0);
            }
            if (a_early) a_early = false;             // (requested ahead of the previous tile's epilogue)
            else if (a_item < item_end) stage_A();    // A rows of the OTHER buffer: released by both groups' LOAD B of the previous K-tile
            stamp(1);
            q_wait_lgkm();
            stamp(2);
            q_bar();
            stamp(3);
            mma(0);
            stamp(4);
            q_bar();
            stamp(5);
            // ---- phase B ----
            if (!abl_noread || kt == 0) ld_a(1);
            int n_new = 0;
            if (b_item < item_end) n_new = stage_B();   // B rows of THIS buffer: both groups have run LOAD A
            stamp(6);
            // K-tile kt+1 complete: all but the pieces just requested -- and, in a tile's first K-tile, the previous tile's epilogue stores, which
            // are younger than the A pieces of K-tile 1 (vmcnt is one in-order counter for loads and stores)
            switch (n_new + pending_stores) {
                case 4: q_wait<4>(); break;
                case 5: q_wait<5>(); break;
                case 20: q_wait<20>(); break;
                case 21: q_wait<21>(); break;
                case 36: q_wait<36>(); break;
                case 37: q_wait<37>(); break;
                default: q_wait<0>(); break;
            }
            pending_stores = 0;
            stamp(7);
            q_bar();
            stamp(8);
            mma(1);
            stamp(9);
            q_bar();
            stamp(10);
            if constexpr (TRACE) {
                if (tr_wg && tr_first && kt == ((p.dbg >> 4) & 15) && lane == 0) {
#pragma unroll
                    for (int i = 0; i < 11; i++) ((unsigned long long*)(lds + Q_LDS))[w * 16 + i] = tr_ts[i];
                }
            }
            cur ^= 1;
        }
        tile_stamp(2);
        if (!STAGGERED_EPI && grp == 0) q_bar();     // let group 1 finish its last MFMA phase: epilogues run together
        const bool inner = (cm0 + QBM <= p.M) && (cn0 + QBN <= p.N);
        // The next tile's K-tile 1 goes into the buffer the last K-tile just left (its A rows: both groups are past LOAD B).  Its A pieces are
        // requested HERE, ahead of the epilogue's stores, so that the counted wait of the next LOAD B can leave the stores in flight.
        // (Only where the epilogue issues a known number of stores and no loads: plain bf16 epilogues on interior tiles -- 2 stores per 32 x 32 tile = 16.)
        // Round 6: also where the quick-GELU epilogue saves its tile for the backward (16 more stores, paired like the output: gemm_common.h); before, that
        // launch waited for vmcnt(4) -- every store acknowledged -- ahead of each tile's first MFMA: fc1 of a layer the backward passes through 362.7 ->
        // 358.0 us as the mean of the step's twelve launches (one with the saved tile), L/14 552.7 -> 509.7 (13 of 24).  Measured and left as they were
        // (profiles/r06_epilogue_waits.md): the same for every other epilogue on interior tiles (f32 outputs, patch embedding, accumulating forms: -0.4 ...
        // -1.6 us per launch; dX through quick-GELU' +4 us: its saved-tile loads queue behind the early A pieces), erf-GELU with its saved tile (+1.8 us).
        // A bias epilogue with aux set is the trace build's stamp buffer.
        constexpr bool PLAIN_EPI = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16);
        constexpr bool AUX_COUNTED = (EPI == EPI_QGELU_BF16);
        if (PLAIN_EPI && inner && (AUX_COUNTED || !p.aux) && a_item < item_end) { stage_A(); a_early = true; pending_stores = (AUX_COUNTED && p.aux) ? 32 : 16; }
        const float* lbias = (const float*)(lds + Q_BIAS_OFF + tile_parity * 1024) + wc * 64;
        // (trace, K-tile selector 15: the stamps record the epilogue instead -- 0 start, 1 bias values in registers, 2..5 row block i converted and its
        //  stores issued, 6 = 5 again)
        auto estamp = [&](int idx) {
            if constexpr (TRACE) { if (tr_wg && tr_first && ((p.dbg >> 4) & 15) == 15) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_ts[idx]) :: "memory"); }
        };
        auto run = [&](auto guard_tag) {
            constexpr bool G = decltype(guard_tag)::value;
            estamp(0);
            // saved pre-activations: a rolling two-row-block prefetch (row blocks i and i + 1 in flight while block i - 1 is converted).  All eight 32 x 32
            // tiles requested up front (64 registers beside the 128 accumulators) spilled 8 VGPRs into scratch.
            constexpr bool AUX_IN = (EPI == EPI_DQGELU_BF16);
            uint4 auxr[AUX_IN ? 2 : 1][2][2];
            if constexpr (AUX_IN && !LINES) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) epi_aux_load<G>(p, cm0 + grp * 128 + i * 32, cn0 + wc * 64 + j * 32, lane, auxr[i][j]);
            }
            // the wave's 32 bias values of this tile, read once (one LDS round trip instead of one per 32 x 32 tile and quad)
            // (not where the epilogue needs the registers itself -- erf-GELU and the activation derivatives spill with 32 more live values)
            constexpr bool BIAS_PRE = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_F32 || EPI == EPI_ACC_F32);
            f32x4 bq[2][4];
            if constexpr (BIAS_PRE && !LINES) { if (has_bias) epi_bias_preload(lbias, hi, bq); }
            if constexpr (LINES && AUX_IN) {
                // dX through quick-GELU' (round 6): the saved derivative tile comes in through the SAME quad-contiguous pattern the stores leave by, two row
                // blocks in flight (block i + 2 requested into the slot block i just left), then one multiply per element
                unsigned ax[2][16];
                epi_lines_aux_load<G>(p, cm0 + grp * 128, cn0 + wc * 64, lane, ax[0]);
                epi_lines_aux_load<G>(p, cm0 + grp * 128 + 32, cn0 + wc * 64, lane, ax[1]);
                estamp(1);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    quad_transpose16(ax[i & 1], lane);
                    epi_lines_bf16<EPI, G>(p, acc[i][0], acc[i][1], cm0 + grp * 128 + i * 32, cn0 + wc * 64, lane, lbias, nullptr, ax[i & 1]);
                    if (i + 2 < 4) epi_lines_aux_load<G>(p, cm0 + grp * 128 + (i + 2) * 32, cn0 + wc * 64, lane, ax[i & 1]);
                    estamp(2 + i);
                }
            } else if constexpr (LINES) {
                f32x4 bl[8];
                if (has_bias) epi_lines_bias_preload(lbias, hi, bl);
                estamp(1);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    epi_lines_bf16<EPI, G>(p, acc[i][0], acc[i][1], cm0 + grp * 128 + i * 32, cn0 + wc * 64, lane, lbias, bl);
                    estamp(2 + i);
                }
            } else {
            estamp(1);
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int64_t mt = cm0 + grp * 128 + i * 32, nt = cn0 + wc * 64 + j * 32;
                    if constexpr (EPI == EPI_F32 || EPI == EPI_ACC_F32) {
                        epi_tile_f32<EPI, G>(p, acc[i][j], mt, nt, lane, lbias + j * 32, BIAS_PRE ? bq[j] : nullptr);
                    } else if constexpr (EPI == EPI_PATCH_F32 || EPI == EPI_PATCHM_F32) {
                        epi_tile_patch<G>(p, acc[i][j], mt, nt, lane);
                    } else {
                        uint4 c0, c1;
                        epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32, AUX_IN ? auxr[AUX_IN ? (i & 1) : 0][j] : nullptr, BIAS_PRE ? bq[j] : nullptr);
                        if constexpr (AUX_IN) { if (i + 2 < 4) epi_aux_load<G>(p, mt + 64, nt, lane, auxr[i & 1][j]); }    // row block i + 2 into the slot just consumed
                        epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                        epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
                    }
                }
                estamp(2 + i);
            }
            }
        };
        if (!abl_noepi) {
            if (inner) run(std::false_type{}); else run(std::true_type{});
        } else {
            pending_stores = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) asm volatile("" :: "v"(acc[i][j]));
        }
        tile_stamp(4);                               // (conversion done, every store of the tile issued)
        if (STAGGERED_EPI) q_bar();                  // the epilogue interval
        tile_stamp(3);
        if constexpr (TRACE) {
            if (tr_wg && tr_first) {
                if (lane == 0 && ((p.dbg >> 4) & 15) == 15) {
#pragma unroll
                    for (int i = 0; i < 11; i++) ((unsigned long long*)(lds + Q_LDS))[w * 16 + i] = tr_ts[i];
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 5; i++) ((unsigned long long*)(lds + Q_LDS))[w * 16 + 11 + i] = tile_ts[i];
                }
                __syncthreads();
                if (threadIdx.x < 128) ((unsigned long long*)p.aux)[threadIdx.x] = ((unsigned long long*)(lds + Q_LDS))[threadIdx.x];
                tr_first = false;
            }
            if (tr_wg && ++tr_tile == (p.dbg >> 8) && !tr_first) tr_first = true;
        }
        item += item_step;
        if (item >= item_end) break;
        tile_parity ^= 1;
    }
    if (STAGGERED_EPI && grp == 0) q_bar();          // group 1's last interval
    if constexpr (TRACE) {          // every workgroup: ticks from its first instruction to its last, and when it started
        unsigned long long wg_t1, wg_r1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1), "=s"(wg_r1) :: "memory");
        if (threadIdx.x == 0) {
            ((unsigned long long*)p.aux)[128 + blockIdx.x] = wg_t1 - wg_t0; ((unsigned long long*)p.aux)[128 + 256 + blockIdx.x] = wg_r0;
            ((unsigned long long*)p.aux)[128 + 512 + blockIdx.x] = wg_r1;
        }
    }
}

#ifdef OWL_TUNING
static int g_pp2_nostore = 0;            // 1: every epilogue store skipped (upper bound on what the store path costs)
OWL_API int owl_gemm_pp2_nostore(int on) { g_pp2_nostore = on; return 0; }
static int g_pp2_slots = 256;            // persistent grid size (tools/: does a GEMM on half the CUs beside the other stream's kernel pay?)
OWL_API int owl_gemm_pp2_slots(int n) { g_pp2_slots = n; return 0; }
static int g_pp2_bw[2] = {0, 0};         // column-block width of the tile order for the bias / quick-GELU epilogue: 0 = the launcher's rule, else the largest divisor of tiles_n up to this
OWL_API int owl_gemm_pp2_block_width(int epi, int bw) { if (epi < 0 || epi > 1) return -1; g_pp2_bw[epi] = bw; return 0; }
static int g_pp2_abl = 0;                // timing-only ablations: bit 0 no LDS-DMA requests after the prologue, bit 1 fragments read once per tile, bit 2 no epilogue
OWL_API int owl_gemm_pp2_ablate(int a) { g_pp2_abl = a; return 0; }
static int g_pp2_lines = 1;              // quad-contiguous stores: 0 off, 1 bias epilogue (the product's choice), 2 quick-GELU epilogue too, 3 + dX through quick-GELU' (loads too)
OWL_API int owl_gemm_pp2_lines(int on) { g_pp2_lines = on; return 0; }
static int g_pp2_stagger = 0;            // start stagger of every second workgroup, in units of s_sleep 127 (~8 k cycles)
OWL_API int owl_gemm_pp2_stagger(int n) { g_pp2_stagger = n; return 0; }
#else
static constexpr int g_pp2_slots = 256;
#endif

template <int EPI>
static int launch_pp2(hipStream_t s, GemmP p) {
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
    });
    p.tiles_m = (int)((p.M + QBM - 1) / QBM); p.tiles_n = (int)((p.N + QBN - 1) / QBN);
    p.dbg = 0;
    p.stagger = 0;
#ifdef OWL_TUNING
    p.dbg = g_pp2_abl << 16;
    p.stagger = g_pp2_stagger;
#endif
    // column-block width of the tile order (see `decode`): the largest divisor of tiles_n up to 4 for the forward epilogues (same-process A/B at
    // M = 73 984: QKV -3 %, half-batch fc1 -4 %, others +-0); the plain row-major order (one block) elsewhere -- dX through quick-GELU' measured
    // 5 % slower blocked (its A panel is then fetched three times, beside the 455 MB of pre-activations it already streams), and so did the
    // WHOLE-batch fc1 inside the model (289 row panels: 0.387 -> 0.363 ms per launch row-major, old / new library alternated in bench.py
    // --encoder-streams 1; the stand-alone tool had it neutral) -- blocked only up to 160 row panels there (the sub-batch launches).
    // Round 3: the whole-batch fc1 (tiles_m > 160: the one-stream schedule) takes blocks of 6 -- per launch 0.353-0.358 ms like row-major (blocks of 4: 0.370, HIP events in
    // bench.py), and the W block (2.4 MB) stays in the XCD's L2: 545 MB requested from the fabric instead of 802 (row-major: 4.7 MB of W per XCD and round, answered by the
    // infinity cache).  At step level every width from 2 to 12 is within 0.1 ms of the others either schedule (tools/tile_order_ab.py); only single columns lose (0.5-1.2 ms).
    p.nsplit = p.tiles_n;
    if (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16)
        for (int d = (EPI == EPI_QGELU_BF16 && p.tiles_m > 160) ? 6 : 4; d >= 1; d--)
            if (p.tiles_n % d == 0) { p.nsplit = d; break; }
#ifdef OWL_TUNING
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16) {
        const int want = g_pp2_bw[EPI == EPI_QGELU_BF16 ? 1 : 0];
        if (want > 0)
            for (int d = want; d >= 1; d--)
                if (p.tiles_n % d == 0) { p.nsplit = d; break; }
    }
#endif
    const int nitems = p.tiles_m * p.tiles_n;
#ifdef OWL_TUNING
    if (g_pp2_nostore) p.M = 0;             // (after the tile counts: every store fails its row guard)
#endif
    p.persistent = nitems > g_pp2_slots ? 1 : 0;
    // Quad-contiguous stores (gemm_common.h, epi_lines_bf16; profiles/r03_gemm_anatomy.md section 2b): the store-bound bias epilogue, -3 ... -5 % at
    // K = 768.  The quick-GELU epilogue is VALU-bound and loses 2-4 % with the transposition on top (tuning builds can still switch it on:
    // owl_gemm_pp2_lines(2)); the erf-GELU one does not fit the register file with it (76 spilled registers).
#ifdef OWL_TUNING
    constexpr bool LINES_OK = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_DQGELU_BF16);
    const bool lines_on = EPI == EPI_BIAS_BF16 ? g_pp2_lines >= 1 : (EPI == EPI_DQGELU_BF16 ? g_pp2_lines >= 3 : g_pp2_lines >= 2);
#else
    // (EPI_DQGELU_BF16 through this epilogue -- saved tile loaded AND result stored quad-contiguous -- needs two row blocks of aux in flight beside the 128
    //  accumulators: 25 spilled VGPRs, and measured +1.3 % (B/16) / +3.8 % (L/14) SLOWER than the accumulator-layout form, bit-identical: tuning builds only,
    //  tools/experiments/dqgelu_lines_ab.py, profiles/r06_dqgelu.md)
    constexpr bool LINES_OK = (EPI == EPI_BIAS_BF16);
    const bool lines_on = true;
#endif
    if constexpr (LINES_OK) {
        if (lines_on && p.N % 8 == 0) {                   // (16-byte pieces: whole pieces inside N)
            static unsigned long long attr_done_l = 0;
            OWL_ONCE_PER_DEVICE(attr_done_l, {
                (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
            });
            hipLaunchKernelGGL((gemm_pp2_kernel<EPI, false, true>), dim3(p.persistent ? g_pp2_slots : nitems), dim3(512), Q_LDS, s, p);
            OWL_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL((gemm_pp2_kernel<EPI>), dim3(p.persistent ? g_pp2_slots : nitems), dim3(512), Q_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

#ifdef OWL_TUNING
static void* g_pp2_trace = nullptr;      // device buffer of 128 x u64: the next bias-epilogue launch runs the stamped kernel (tools/pp2_trace.py)
static int g_pp2_trace_tile = 0, g_pp2_trace_kt = 4;
OWL_API int owl_gemm_pp2_trace(void* buf) { g_pp2_trace = buf; return 0; }
OWL_API int owl_gemm_pp2_trace_tile(int n) { g_pp2_trace_tile = n; return 0; }
OWL_API int owl_gemm_pp2_trace_ktile(int n) { g_pp2_trace_kt = n; return 0; }
#endif

// called from gemm.hip's dispatcher; returns 1 if this variant does not handle `epi`
int owl_gemm_pp2_launch(hipStream_t s, int epi, const GemmP& p) {
#ifdef OWL_TUNING
    if (g_pp2_trace && epi == EPI_BIAS_BF16) {
        GemmP q = p;
        q.aux = nullptr;
        (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI_BIAS_BF16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS + 2048);
        q.tiles_m = (int)((q.M + QBM - 1) / QBM); q.tiles_n = (int)((q.N + QBN - 1) / QBN); q.dbg = (g_pp2_trace_tile << 8) | ((g_pp2_trace_kt & 15) << 4); q.nsplit = q.tiles_n;
        for (int d = 4; d >= 1; d--) if (q.tiles_n % d == 0) { q.nsplit = d; break; }
        const int nitems = q.tiles_m * q.tiles_n;
        q.persistent = nitems > 256 ? 1 : 0;
        GemmP qq = q; qq.aux = g_pp2_trace;           // (the bias epilogue never reads aux; PLAIN_EPI's early A staging is off with aux set -- fine for a trace)
        if (g_pp2_lines) {
            (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI_BIAS_BF16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS + 2048);
            hipLaunchKernelGGL((gemm_pp2_kernel<EPI_BIAS_BF16, true, true>), dim3(q.persistent ? 256 : nitems), dim3(512), Q_LDS + 2048, s, qq);
            return 0;
        }
        hipLaunchKernelGGL((gemm_pp2_kernel<EPI_BIAS_BF16, true>), dim3(q.persistent ? 256 : nitems), dim3(512), Q_LDS + 2048, s, qq);
        return 0;
    }
#endif
    switch (epi) {
        case EPI_BIAS_BF16: return launch_pp2<EPI_BIAS_BF16>(s, p);
        case EPI_QGELU_BF16: return launch_pp2<EPI_QGELU_BF16>(s, p);
        case EPI_DQGELU_BF16: return launch_pp2<EPI_DQGELU_BF16>(s, p);
        case EPI_GELU_BF16: return launch_pp2<EPI_GELU_BF16>(s, p);
        case EPI_DGELU_BF16: return launch_pp2<EPI_DGELU_BF16>(s, p);
        case EPI_F32: return launch_pp2<EPI_F32>(s, p);               // class head e = W feats + b, dfeats
        case EPI_ACC_F32: return launch_pp2<EPI_ACC_F32>(s, p);       // dfeats += (box head)
        case EPI_PATCH_F32: return launch_pp2<EPI_PATCH_F32>(s, p);   // patch embedding, A gathered from the image
        case EPI_PATCHM_F32: return launch_pp2<EPI_PATCHM_F32>(s, p); // ... from an explicit im2row matrix (L/14)
        default: return 1;                                            // (the transposing epilogue stays on the four-phase kernel)
    }
}
