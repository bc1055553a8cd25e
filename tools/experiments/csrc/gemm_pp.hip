// Ping-pong ("8-phase") variant of the 256x256x64 bf16 GEMM for the bf16-output epilogues.
//
// Same tile, same LDS image (LDS-DMA, 16-byte chunk XOR swizzle), same epilogues as gemm.hip's kernel -- what changes is
// the SCHEDULE.  The 8 waves form two groups (waves 0-3 = output rows 0-127, waves 4-7 = rows 128-255; every SIMD hosts one
// wave of each).  A K-tile is processed as four quadrant phases (64x32 of the wave's 128x64 output, K = 64: 8 MFMAs of
// 32x32x16), and every phase is split by barriers into a LOAD half (ds_reads of the quadrant's fragments + LDS-DMA staging)
// and an MFMA half.  Group 1 runs one barrier behind group 0, so in every half-slot exactly one group owns the matrix pipe
// while the other one waits for LDS / issues DMA -- instead of both stalling on their ds_reads and then fighting for the pipe.
//
//   half-slot:     2p            2p+1          2p+2          2p+3
//   group 0:   LOAD  q(p)     MFMA q(p)     LOAD  q(p+1)   MFMA q(p+1)
//   group 1:   MFMA q(p-1)    LOAD  q(p)    MFMA q(p)      LOAD  q(p+1)
//
// Quadrant order per K-tile: (i0,j0) (i0,j1) (i1,j1) (i1,j0); A(i0) + B(j0) are read in q0, B(j1) in q1, A(i1) in q2 (same
// registers as A(i0)), nothing in q3: both B fragments stay live for the K-tile.  Consequently the K-tile's LDS half-tiles
// become free EARLY: B rows after both groups' q1 LOAD, A rows 0-127 (group 0 only) after group 0's q2 LOAD, A rows 128-255
// after group 1's q2 LOAD -- and K-tile c+2 is staged into the buffer K-tile c is still being computed from:
//     q2 LOAD: B half-tiles of K-tile c+2 (4 DMA instructions per wave, + the bias slice when c+2 opens a tile)
//     q3 LOAD: A half-tiles of K-tile c+2 (4), then s_waitcnt vmcnt(N): retire K-tile c+1, leave c+2 in flight
// i.e. every DMA has more than a full K-tile of MFMA time to land, and the waits are counted (vmcnt is one in-order
// counter for loads AND stores on gfx950; N includes the previous tile's epilogue stores right after a tile boundary).
// Every LOAD half ends with lgkmcnt(0) BEFORE its barrier: a ds_read still in flight at a raw s_barrier can lose the race
// against another wave's post-barrier LDS-DMA (found by the determinism test on the first kernel).
//
// At a tile boundary group 0 waits one extra barrier (group 1's last MFMA half) so that both groups run their epilogues
// together; group 1's extra barrier at the next tile start re-creates the offset.  Barrier counts per tile: 8 nk + 1 each.
#ifdef OWL_TUNING   // whole file: a tuning-build experiment, not part of the shipped library
#include "gemm_common.h"
#include <type_traits>

static constexpr int PBM = 256, PBN = 256, PBK = 64;
static constexpr int P_A_BYTES = PBM * PBK * 2, P_B_BYTES = PBN * PBK * 2, P_STAGE = P_A_BYTES + P_B_BYTES;   // 32 + 32 KiB
static constexpr int P_BIAS_OFF = 2 * P_STAGE, P_LDS = 2 * P_STAGE + 2 * 1024;

template <int N> __device__ __forceinline__ void pp_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pp_bar() {      // sched_barrier: the scheduler must not move MFMAs / ds_reads across the half-slot boundary
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// TRACE (tuning only, tools/pp_trace.py): the first workgroup stamps s_memtime at every half-slot boundary of K-tile 4
// of its first tile into 4 KiB of LDS behind the regular image and dumps them to p.aux at the end.
template <int EPI, bool TRACE = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool TRANS = (EPI == EPI_TRANS_BF16);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int grp = w >> 2, wc = w & 3;              // group = output-row half; wc = 64-column slice
    const int nk = (int)(p.K / PBK);                 // >= 2 (host checks)
    const int nitems = p.tiles_m * p.tiles_n;
    int item, item_end, item_step;
    if (p.persistent) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, chunk = (nitems + 7) >> 3;
        item = xcd * chunk + idx; item_end = min(nitems, (xcd + 1) * chunk); item_step = gridDim.x >> 3;
    } else {
        item = xcd_remap(blockIdx.x, nitems); item_end = item + 1; item_step = 1;
    }
    if (item >= item_end) return;

    // ---- DMA stream: K-tiles in consumption order across the persistent tile loop --------------------------------
    // wave w stages rows half*128 + (w*2+q)*8 + (lane>>3), q = 0,1, of every half-tile (16 x 1 KiB instructions / 8 waves)
    int s_item = item, s_k = 0, s_buf = 0, s_parity = 0;
    // per-lane byte offsets relative to the tile's (wave-uniform) base pointers: 8 VGPRs instead of 8 64-bit pointers
    unsigned a_voff[2][2], w_voff[2][2], b_voff = 0;   // [half][q]
    // (buffer_load ... lds with a scalar K offset, which pays in the attention kernels, measured 2-3 % SLOWER here: the
    // ~57-cycle issue cost of an LDS-DMA piece does not depend on the address form, tools/pp_trace.py)
    const bf16_t* a_base = nullptr;
    const bf16_t* w_base = nullptr;
    const float* b_base = nullptr;
    auto stream_setup = [&]() {
        const int tm = s_item / p.tiles_n, tn = s_item - tm * p.tiles_n;
        const int64_t m0 = (int64_t)tm * PBM, n0 = (int64_t)tn * PBN;
        a_base = p.A + m0 * p.lda;
        w_base = p.W + n0 * p.ldw;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int r = h * 128 + (w * 2 + q) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int64_t am = m0 + r; if (am >= p.a_rows) am = p.a_rows - 1;
                int64_t wn = n0 + r; if (wn >= p.w_rows) wn = p.w_rows - 1;
                a_voff[h][q] = (unsigned)(((am - m0) * p.lda + c * 8) * 2);
                w_voff[h][q] = (unsigned)(((wn - n0) * p.ldw + c * 8) * 2);
            }
        if (p.bias) {
            int64_t n = n0 + lane * 4; if (n + 4 > p.N) n = p.N - 4;
            b_base = p.bias + n0;
            b_voff = (unsigned)((n - n0) * 4);
        }
    };
    const bool has_bias = p.bias != nullptr;
    auto stream_live = [&]() { return s_item < item_end; };
    // B half-tiles (+ bias slice when the K-tile opens a tile): 4 or 5 VMEM ops
    auto stage_B = [&]() -> int {
        if (s_k == 0) stream_setup();
        unsigned char* base = lds + s_buf * P_STAGE + P_A_BYTES;
        const unsigned char* g = (const unsigned char*)(w_base + (int64_t)s_k * PBK);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + w_voff[h][q]), LPTR(base + (h * 128 + (w * 2 + q) * 8) * 128), 16, 0, 0);
        if (s_k == 0 && has_bias) {
            __builtin_amdgcn_global_load_lds(GPTR((const unsigned char*)b_base + b_voff), LPTR(lds + P_BIAS_OFF + s_parity * 1024), 16, 0, 0);
            return 5;
        }
        return 4;
    };
    // A half-tiles: 4 VMEM ops; advances the stream
    auto stage_A = [&]() {
        unsigned char* base = lds + s_buf * P_STAGE;
        const unsigned char* g = (const unsigned char*)(a_base + (int64_t)s_k * PBK);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + a_voff[h][q]), LPTR(base + (h * 128 + (w * 2 + q) * 8) * 128), 16, 0, 0);
        s_buf ^= 1;
        if (++s_k == nk) { s_k = 0; s_item += item_step; s_parity ^= 1; }
    };

    // fragment addresses: A rows grp*128 + i*32 + (lane&31), B rows wc*64 + j*32 + (lane&31); the swizzle term
    // ((row>>1)&7) is the same for every 32-row tile of a lane, so one base + one swizzle per operand suffices
    const int a_row0 = grp * 128 + (lane & 31), b_row0 = wc * 64 + (lane & 31);
    const int a_base_off = a_row0 * 128, b_base_off = P_A_BYTES + b_row0 * 128;
    const int a_swz = (a_row0 >> 1) & 7, b_swz = (b_row0 >> 1) & 7;

    // prologue: K-tiles 0 (+bias) and 1 in flight, retire K-tile 0
    stage_B(); stage_A();
    if (stream_live()) { stage_B(); stage_A(); pp_wait<8>(); } else { pp_wait<0>(); }
    pp_bar();

    int cur = 0, tile_parity = 0;
    const bool tr_wg = TRACE && blockIdx.x == 0;
    bool tr_first = true;
    unsigned long long tr_ts[20] = {};
    int pending_stores = 0;          // epilogue stores issued after the newest in-flight K-tile's DMA (0 or 16)
    bool deferred_A = false;         // TRANS: the A pieces of the K-tile after next wait for the epilogue to release its staging area
    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        const int tm = item / p.tiles_n, tn = item - tm * p.tiles_n;
        const int64_t cm0 = (int64_t)tm * PBM, cn0 = (int64_t)tn * PBN;
        if (grp == 1) pp_bar();                       // (re-)create the one-barrier offset
        for (int kt = 0; kt < nk; kt++) {
            const unsigned char* tb = lds + cur * P_STAGE;
            // s_memtime is issued WITHOUT waiting for its result (a waited stamp costs ~140 cycles and distorts everything);
            // the 20 results of K-tile 4 stay in SGPRs until the K-tile's last barrier
            auto stamp = [&](int idx) {
                if constexpr (TRACE) {
                    if (tr_wg && tr_first && kt == 4) {
                        // (the last stamps are read back right away: those wait for their result)
                        if (idx >= 17) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_ts[idx]));
                        else asm volatile("s_memtime %0" : "=s"(tr_ts[idx]));
                    }
                }
            };
            bf16x8 fa[2][4], fb[2][4];                // [tile within quadrant][kc], [j][kc]
            auto ld_a = [&](int ih) {
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int kc = 0; kc < 4; kc++)
                        fa[t][kc] = *(const bf16x8*)(tb + a_base_off + (2 * ih + t) * 4096 + (((kc * 2 + hi) ^ a_swz) << 4));
            };
            auto ld_b = [&](int j) {
#pragma unroll
                for (int kc = 0; kc < 4; kc++) fb[j][kc] = *(const bf16x8*)(tb + b_base_off + j * 4096 + (((kc * 2 + hi) ^ b_swz) << 4));
            };
            auto mma = [&](int ih, int j) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        if constexpr (TRANS) acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][kc], fb[j][kc], acc[2 * ih + t][j], 0, 0, 0);
                        else acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kc], fa[t][kc], acc[2 * ih + t][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
            };
            // ---- q0 ----
            stamp(0);
            ld_b(0); ld_a(0);
            stamp(1); pp_wait_lgkm(); stamp(2); pp_bar(); stamp(3);
            mma(0, 0);
            stamp(4); pp_bar();
            // ---- q1 ----
            stamp(5);
            ld_b(1);
            stamp(6); pp_wait_lgkm(); stamp(7); pp_bar(); stamp(8);
            mma(0, 1);
            stamp(9); pp_bar();
            // ---- q2: B rows of this buffer are free (both groups finished their q1 LOAD) ----
            stamp(10);
            ld_a(1);
            int n_new = 0;
            const bool live = stream_live();
            if (live) n_new = stage_B();
            stamp(11); pp_wait_lgkm(); stamp(12); pp_bar(); stamp(13);
            mma(1, 1);
            stamp(14); pp_bar();
            // ---- q3: A rows are free; retire K-tile kt+1, leave kt+2 in flight.  The transposing epilogue borrows this
            //      buffer's A rows as its staging area, so at a tile's last K-tile it defers the A pieces until after it ----
            stamp(15);
            if (live && !(TRANS && kt + 1 == nk)) { stage_A(); n_new += 4; }
            else if (live) deferred_A = true;
            stamp(16);
            {
                const int total = n_new + pending_stores;
                if (total == 8) pp_wait<8>();
                else if (total == 9) pp_wait<9>();
                else if (total == 24) pp_wait<24>();
                else if (total == 25) pp_wait<25>();
                else if (total == 4) pp_wait<4>();
                else if (total == 5) pp_wait<5>();
                else if (total == 20) pp_wait<20>();
                else if (total == 21) pp_wait<21>();
                else pp_wait<0>();
                pending_stores = 0;
            }
            stamp(17); pp_bar(); stamp(18);
            mma(1, 0);
            stamp(19); pp_bar();
            if constexpr (TRACE) {
                if (tr_wg && tr_first && kt == 4) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 20; i++) ((unsigned long long*)(lds + P_LDS))[w * 40 + i] = tr_ts[i];
                    }
                }
            }
            cur ^= 1;
        }
        if (grp == 0) pp_bar();                       // let group 1 finish its last MFMA half: epilogues run together
        const bool inner = (cm0 + PBM <= p.M) && (cn0 + PBN <= p.N);
        if constexpr (TRANS) {
            // Per-head transposed store out_t[b][n][t]: written straight from the accumulators a lane's 16-byte pieces land
            // in 32 different 64-byte sectors per instruction (WRITE_SIZE 1.45x the output).  Instead each wave stages a
            // [32 n][64 t] bf16 image (16-byte chunks XOR-swizzled by n&7) in its 4 KiB of the A rows this tile no longer
            // needs and writes it back as 8 rows x 128 contiguous bytes per instruction.
            unsigned char* img = lds + s_buf * P_STAGE + w * 4096;
            const float* lbias = (const float*)(lds + P_BIAS_OFF + tile_parity * 1024) + wc * 64;
            const int n_l = lane & 31;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float bv = 0.f;
                if (p.bias) bv = lds_read_f1(lbias + j * 32 + n_l);
#pragma unroll
                for (int ip = 0; ip < 2; ip++) {
                    // accumulator register r of tile ti: token 8*(r>>2) + 4*hi + (r&3) of its 32, column n_l
#pragma unroll
                    for (int ti = 0; ti < 2; ti++)
#pragma unroll
                        for (int qd = 0; qd < 4; qd++) {
                            const f32x16& a = acc[2 * ip + ti][j];
                            uint2 v;
                            v.x = pack_bf2(a[qd * 4 + 0] + bv, a[qd * 4 + 1] + bv);
                            v.y = pack_bf2(a[qd * 4 + 2] + bv, a[qd * 4 + 3] + bv);
                            *(uint2*)(img + n_l * 128 + (((ti * 4 + qd) ^ (n_l & 7)) << 4) + hi * 8) = v;
                        }
                    const int64_t m_base = cm0 + grp * 128 + ip * 64;
                    const int64_t n_base = cn0 + wc * 64 + j * 32;
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int nr = it * 8 + (lane >> 3), c16 = lane & 7;
                        const f32x4 raw = lds_read_f4(img + nr * 128 + ((c16 ^ (nr & 7)) << 4));
                        const int64_t n = n_base + nr, m = m_base + c16 * 8;
                        if (n < p.N && m < p.M) {
                            const int64_t bimg = m / p.Tp, t = m - bimg * p.Tp;
                            *(f32x4*)((bf16_t*)p.out + (bimg * p.N + n) * p.Tp + t) = raw;
                        }
                    }
                }
            }
            // the staging slices are wave-private, but the deferred DMA below is partitioned differently: everybody must be done
            pp_wait_lgkm();
            pp_bar();
            if (deferred_A) { stage_A(); deferred_A = false; }
        } else {
            const float* lbias = (const float*)(lds + P_BIAS_OFF + tile_parity * 1024) + wc * 64;
            auto run = [&](auto guard_tag) {
                constexpr bool G = decltype(guard_tag)::value;
                constexpr bool AUX_IN = (EPI == EPI_DQGELU_BF16);     // (the erf-GELU derivative needs the registers itself: per-tile loads there)
                // the saved pre-activations of all eight 32x32 tiles are requested up front (64 registers: the K loop's fragments are
                // dead here): one HBM latency per 256x256 tile instead of one per 32x32 tile -- no load is ordered behind a store
                uint4 auxr[AUX_IN ? 4 : 1][2][2];
                if constexpr (AUX_IN) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) epi_aux_load<G>(p, cm0 + grp * 128 + i * 32, cn0 + wc * 64 + j * 32, lane, auxr[i][j]);
                }
                // the wave's bias values, one LDS round trip (gemm_common.h, epi_bias_preload) -- not where the epilogue needs the registers itself
                constexpr bool BIAS_PRE = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_F32 || EPI == EPI_ACC_F32);
                f32x4 bq[2][4];
                if constexpr (BIAS_PRE) { if (p.bias) epi_bias_preload(lbias, lane >> 5, bq); }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int64_t mt = cm0 + grp * 128 + i * 32, nt = cn0 + wc * 64 + j * 32;
                        if constexpr (EPI == EPI_F32 || EPI == EPI_ACC_F32) {
                            epi_tile_f32<EPI, G>(p, acc[i][j], mt, nt, lane, lbias + j * 32, BIAS_PRE ? bq[j] : nullptr);
                        } else {
                            uint4 c0, c1;
                            epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32, AUX_IN ? auxr[AUX_IN ? i : 0][j] : nullptr, BIAS_PRE ? bq[j] : nullptr);
                            epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                            epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
                        }
                    }
            };
            if (inner) run(std::false_type{}); else run(std::true_type{});
        }
        if constexpr (TRACE) {
            if (tr_wg && tr_first) {
                __syncthreads();
                if (threadIdx.x < 320) ((unsigned long long*)p.aux)[threadIdx.x] = ((unsigned long long*)(lds + P_LDS))[threadIdx.x];
                tr_first = false;
            }
        }
        item += item_step;
        if (item >= item_end) break;
        tile_parity ^= 1;
        // counted wait across the epilogue is only valid when this wave really issued its 16 stores (plain inner tiles)
        pending_stores = (inner && (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16) && !p.aux) ? 16 : 0;    // (TRANS: its stores precede the deferred A pieces)
        if (pending_stores == 0 && !TRANS) pp_wait<0>();   // otherwise drain now: later counted waits assume nothing unknown is pending
                                                             // (TRANS: the stores are older than the deferred A pieces every later wait retires)
    }
}

template <int EPI>
static int launch_pp(hipStream_t s, GemmP p, int slots_override, int persistent_on, int nostore) {
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    });
    p.tiles_m = (int)((p.M + PBM - 1) / PBM); p.tiles_n = (int)((p.N + PBN - 1) / PBN);
    p.nsplit = 1;
    if (nostore == 1) p.M = 0;
    p.dbg = nostore & ~1;
    const int nitems = p.tiles_m * p.tiles_n;
    const int slots = slots_override ? slots_override : 256;
    p.persistent = (persistent_on && nitems > slots) ? 1 : 0;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI>), dim3(p.persistent ? slots : nitems), dim3(512), P_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

// called from gemm.hip's dispatcher; returns 1 if this variant does not handle `epi`
int owl_gemm_pp_launch(hipStream_t s, int epi, const GemmP& p, int slots_override, int persistent_on, int nostore) {
    switch (epi) {
        case EPI_BIAS_BF16:
            if (nostore & 8) {                       // trace run (p.aux = 320 x u64 trace buffer)
                GemmP q = p;
                (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI_BIAS_BF16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS + 4096);
                q.tiles_m = (int)((q.M + PBM - 1) / PBM); q.tiles_n = (int)((q.N + PBN - 1) / PBN); q.nsplit = 1; q.dbg = 0;
                const int nitems = q.tiles_m * q.tiles_n;
                q.persistent = nitems > 256 ? 1 : 0;
                hipLaunchKernelGGL((gemm_pp_kernel<EPI_BIAS_BF16, true>), dim3(q.persistent ? 256 : nitems), dim3(512), P_LDS + 4096, s, q);
                return 0;
            }
            return launch_pp<EPI_BIAS_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_QGELU_BF16: return launch_pp<EPI_QGELU_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_TRANS_BF16: return launch_pp<EPI_TRANS_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_DQGELU_BF16: return launch_pp<EPI_DQGELU_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_GELU_BF16: return launch_pp<EPI_GELU_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_DGELU_BF16: return launch_pp<EPI_DGELU_BF16>(s, p, slots_override, persistent_on, nostore);
        case EPI_F32: return launch_pp<EPI_F32>(s, p, slots_override, persistent_on, nostore);           // class head e = W feats + b, dfeats
        case EPI_ACC_F32: return launch_pp<EPI_ACC_F32>(s, p, slots_override, persistent_on, nostore);   // dfeats += (box head)
        default: return 1;
    }
}
#endif  // OWL_TUNING (whole file)
