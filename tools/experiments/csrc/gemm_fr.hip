// "Free-running" bf16 GEMM: TWO independent four-wave workgroups per CU, a 128 x 256 output tile each, BK = 32, three LDS stages.
//
// Why: in the 256 x 256 ping-pong kernels (gemm_pp2.hip) both wave groups of the one workgroup share the B tiles and every barrier, so a tile's
// epilogue (bias: ~15 %, quick-GELU: ~21 % of a K = 768 tile, profiles/r03_gemm_anatomy.md sections 1 / 5) and the ramp of its first K-tile are
// intervals in which the CU's matrix pipe idles.  Here a CU hosts two workgroups that share nothing -- no LDS, no barrier -- so one's epilogue, tile
// ramp and LOAD work run under the other's MFMAs with no code coupling them; each SIMD holds one wave of each.
//   tile 128 x 256, four waves, wave wc = all 128 rows x the 64-column slice wc: acc[4][2] of 32 x 32 -- exactly one wave GROUP of gemm_pp2.hip, hence
//   the same per-wave epilogue code (gemm_common.h) and, K-steps consumed in order on the same MFMA shape, the same bits as every other kernel here;
//   BK = 32: a stage = A 128 x 32 (8 KiB) + B 256 x 32 (16 KiB); three stages + two bias slices = 74 KiB per workgroup, 148 of the CU's 160 KiB;
//   LDS image: 64-byte rows, the four 16-byte chunks of a row XOR-swizzled with ((row >> 2) & 3) on the DMA source and on the fragment reads
//   (conflict-free for ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}: their row quads carry four different keys);
//   one barrier per K-step:
//       wait F0 | read F1 = fragments (s, k16 step 1) | 8 MFMAs on F0 | wait F1 + this wave's pieces of stage s+1 (counted vmcnt) | BARRIER |
//       request stage s+3 into the buffer of stage s (everybody is past its last read of it) | read F0 = fragments (s+1, k16 step 0) | 8 MFMAs on F1
//   so fragment reads run half a K-step ahead of the MFMAs that use them (two 6-fragment register sets), a DMA piece has two K-steps of flight, and the
//   stream of requests runs on across tile boundaries of the persistent loop (one cursor).
// The price: B is no longer shared by 256 rows -- 1.5x the LDS-DMA bytes per FLOP of the 256 x 256 tile on the CU's 64 B / clk L1 path.
//
// RESULT (profiles/r04_gemm_fr.md): bit-identical to every other kernel on the first run -- and 18-45 % SLOWER than what ships (QKV 303 vs 234 us, fc1 469 vs
// 350, fc2 420 vs 290).  The ablations say why: without its LDS-DMA requests the same loop runs 13-22 % FASTER than the shipped kernel (the epilogue does hide),
// the requests cost 100-170 us whether they are waited for or not and however they are spread -- a request costs by the cache lines it touches, BK = 32 rows are
// half lines, and at this kernel's 16 MFMAs per 6 requests the bare half-line stream reaches 9.4 TB/s / 900 TFLOP/s (tools/probe/lds_dma_rate.hip): the kernel
// sits on that rate.  Re-pointed at whole lines (a BK = 64 image) it is still 15-20 % behind: 1.5x the staging bytes per FLOP cost more than the hidden epilogue
// returns.  Not shipped; kept for OWL_TUNING builds (tile = 5, tools/gemm_fr_bench.py, tools/gemm_fr_ablate.py).
#ifdef OWL_TUNING
#include "gemm_common.h"
#include <type_traits>

static constexpr int FBM = 128, FBN = 256, FBK = 32;
static constexpr int F_A_BYTES = FBM * FBK * 2, F_B_BYTES = FBN * FBK * 2, F_STAGE = F_A_BYTES + F_B_BYTES;   // 8 + 16 KiB
static constexpr int F_NSTAGE = 3, F_BIAS_OFF = F_NSTAGE * F_STAGE, F_LDS = F_BIAS_OFF + 2 * 1024;            // 75 776 B

template <int N> __device__ __forceinline__ void f_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void f_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void f_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// hand-issued fragment read: invisible to hipcc's "LDS read after an LDS-DMA needs vmcnt(0)" rule; the caller waits (f_wait / f_wait_lgkm)
__device__ __forceinline__ bf16x8 f_ldsr(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// LINES: the W tile is staged with its rows permuted inside every 64-row group so that a lane's accumulators are 64 contiguous output bytes, and the
// epilogue stores quad-contiguous (gemm_common.h, epi_lines_bf16): the bias epilogue.
// ABL (OWL_TUNING builds, timing only -- results are wrong): bit 0 = no LDS-DMA requests after the prologue, bit 1 = fragments read once per tile;
// 16 = (timing only) the same bytes requested as whole-line pieces (8 rows x 128 B); 8 = requests issued but never waited for (timing only); 4 = (correct results) the stage's six requests spread over the second MFMA half, one behind each of its first six MFMAs, instead of a burst behind the barrier
template <int EPI, bool LINES, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_fr_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wc = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int nk = (int)(p.K / FBK);                 // >= 3 (host checks)
    const int nitems = p.tiles_m * p.tiles_n;
    const int bw = p.nsplit;                         // column-block width of the tile order (as gemm_pp2.hip)
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

    // ---- the DMA cursor: K-steps in consumption order across the persistent tile loop -------------------------------------------------
    int c_item = item, c_k = 0, c_buf = 0, c_parity = 0;
    unsigned a_voff[2], w_voff[4], b_voff = 0;       // wave wc stages A rows 32 wc + 16 q + (lane >> 2), W rows 64 wc + 16 q + (lane >> 2)
    const bf16_t* a_base = nullptr;
    const bf16_t* w_base = nullptr;
    const float* b_base = nullptr;
    const bool has_bias = p.bias != nullptr;
    bool st_first = false;
    const unsigned char* st_ga = nullptr; const unsigned char* st_gw = nullptr; unsigned char* st_base = nullptr;
    const unsigned char* st_gp[6] = {};              // (ABL & 4: the six per-lane source addresses, formed before the MFMAs they are issued between)
    auto stage_prep = [&]() {
        const bool first = c_k == 0;
        st_first = first;
        if (first) {
            int tm, tn; decode(c_item, tm, tn);
            const int64_t m0 = (int64_t)tm * FBM, n0 = (int64_t)tn * FBN;
            a_base = p.A + m0 * p.lda;
            w_base = p.W + n0 * p.ldw;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int r = 32 * wc + 16 * q + (lane >> 2);
                const int c = (lane & 3) ^ ((r >> 2) & 3);
                int64_t am = m0 + r; if (am >= p.a_rows) am = p.a_rows - 1;
                a_voff[q] = (unsigned)(((am - m0) * p.lda + c * 8) * 2);
                if constexpr ((ABL & 16) != 0) {       // (timing only: the same bytes as WHOLE-line requests, 8 rows x 128 B -- what a BK = 64 image would cost)
                    const int r8 = 32 * wc + 8 * q + (lane >> 3);
                    a_voff[q] = (unsigned)((r8 * p.lda + (lane & 7) * 8) * 2);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 64 * wc + 16 * q + (lane >> 2);
                const int c = (lane & 3) ^ ((r >> 2) & 3);
                // LINES: LDS row (j, qd, hi, e) of a 64-row group <- W row 32 hi + 16 j + 4 qd + e of the group (the XOR key stays the LDS row's)
                const int rs = LINES ? ((r & ~63) | (((r >> 2) & 1) << 5) | (((r >> 5) & 1) << 4) | (((r >> 3) & 3) << 2) | (r & 3)) : r;
                int64_t wn = n0 + rs; if (wn >= p.w_rows) wn = p.w_rows - 1;
                w_voff[q] = (unsigned)(((wn - n0) * p.ldw + c * 8) * 2);
                if constexpr ((ABL & 16) != 0) {
                    const int r8 = 64 * wc + 8 * q + (lane >> 3);
                    w_voff[q] = (unsigned)((r8 * p.ldw + (lane & 7) * 8) * 2);
                }
            }
            if (has_bias) {
                int64_t n = n0 + lane * 4; if (n + 4 > p.N) n = p.N - 4;
                b_base = p.bias + n0;
                b_voff = (unsigned)((n - n0) * 4);
            }
        }
        st_base = lds + c_buf * F_STAGE;
        st_ga = (const unsigned char*)(a_base + (int64_t)c_k * FBK);
        st_gw = (const unsigned char*)(w_base + (int64_t)c_k * FBK);
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
            for (int q = 0; q < 2; q++) st_gp[q] = st_ga + a_voff[q];
#pragma unroll
            for (int q = 0; q < 4; q++) st_gp[2 + q] = st_gw + w_voff[q];
#pragma unroll
            for (int q = 0; q < 6; q++) asm volatile("" : "+v"(st_gp[q]));
        }
    };
    auto stage_issue = [&](int q) {                  // request q of the stage: 0, 1 = A pieces, 2 .. 5 = W pieces
        if constexpr ((ABL & 4) != 0) {
            __builtin_amdgcn_global_load_lds(GPTR(st_gp[q]), LPTR(st_base + (q < 2 ? (2 * wc + q) * 1024 : F_A_BYTES + (4 * wc + q - 2) * 1024)), 16, 0, 0);
            return;
        }
        if (q < 2) __builtin_amdgcn_global_load_lds(GPTR(st_ga + a_voff[q]), LPTR(st_base + (2 * wc + q) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(GPTR(st_gw + w_voff[q - 2]), LPTR(st_base + F_A_BYTES + (4 * wc + q - 2) * 1024), 16, 0, 0);
    };
    auto stage_finish = [&]() -> int {
        int n_ops = 6;
        if (st_first && has_bias) {
            __builtin_amdgcn_global_load_lds(GPTR((const unsigned char*)b_base + b_voff), LPTR(lds + F_BIAS_OFF + c_parity * 1024), 16, 0, 0);
            n_ops = 7;
        }
        c_buf = c_buf == F_NSTAGE - 1 ? 0 : c_buf + 1;
        if (++c_k == nk) { c_k = 0; c_item += item_step; c_parity ^= 1; }
        return n_ops;
    };
    auto stage = [&]() -> int {                      // 6 VMEM ops (7 with the tile's bias slice)
        stage_prep();
#pragma unroll
        for (int q = 0; q < 6; q++) stage_issue(q);
        return stage_finish();
    };

    // fragment addresses (absolute 32-bit LDS addresses; row = 32 i + (lane & 31): the swizzle key depends on the lane only)
    const int r31 = lane & 31, swz = (r31 >> 2) & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    unsigned fa_off[2], fb_off[2];                   // [k16 step]
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        fa_off[kc] = lds0 + r31 * 64 + (((kc * 2 + hi) ^ swz) << 4);
        fb_off[kc] = lds0 + F_A_BYTES + (wc * 64 + r31) * 64 + (((kc * 2 + hi) ^ swz) << 4);
    }

    // prologue: stages 0, 1, 2 requested; stage 0 landed
    const int n0 = stage();
    int n1 = 0, n2 = 0;
    if (c_item < item_end) n1 = stage();
    if (c_item < item_end) n2 = stage();
    (void)n0;
    switch (n1 + n2) {
        case 12: f_wait<12>(); break;
        case 13: f_wait<13>(); break;
        default: f_wait<0>(); break;
    }
    f_bar();

    int cur = 0, tile_parity = 0;
    int store_age = 0;               // K-steps for which the previous epilogue's 16 stores are still YOUNGER than the stage the counted wait retires
    int n_last = n2;                 // VMEM ops of the newest request
    constexpr bool PLAIN_EPI = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16);
    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        int tm, tn; decode(item, tm, tn);
        const int64_t cm0 = (int64_t)tm * FBM, cn0 = (int64_t)tn * FBN;
        bf16x8 fa[2][4], fb[2][2];                   // [k16 step][row tile i], [k16 step][column tile j]
        auto rd = [&](int kc, int buf) {             // 6 fragment reads of stage `buf`
            const unsigned sb = (unsigned)(buf * F_STAGE);
#pragma unroll
            for (int j = 0; j < 2; j++) fb[kc][j] = f_ldsr(fb_off[kc] + sb + j * 2048);
#pragma unroll
            for (int i = 0; i < 4; i++) fa[kc][i] = f_ldsr(fa_off[kc] + sb + i * 2048);
        };
        auto mma = [&](int kc) {                     // 8 MFMAs, eight independent chains
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kc][j], fa[kc][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        rd(0, cur);
        for (int kt = 0; kt < nk; kt++) {
            const int nxt = cur == F_NSTAGE - 1 ? 0 : cur + 1;
            f_wait_lgkm();
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 2) || kt == 0) rd(1, cur);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            // F1 in registers; this wave's pieces of the NEXT stage landed: everything but the newest request -- and, for two K-steps after an
            // epilogue, its stores, which are younger than that stage's pieces (vmcnt is one in-order counter for loads and stores)
            if constexpr ((ABL & 8) != 0) f_wait_lgkm();          // (ablation: requests issued, never waited for)
            else
            switch (n_last + (store_age > 0 ? 16 : 0)) {
                case 6: f_wait<6>(); break;
                case 7: f_wait<7>(); break;
                case 22: f_wait<22>(); break;
                case 23: f_wait<23>(); break;
                default: f_wait<0>(); break;
            }
            if (store_age > 0) store_age--;
            f_bar();
            // every wave is past its last read of stage `cur`: it takes the request three K-steps ahead
            n_last = 0;
            if (ABL & 1) {              // (ablation: the cursor advances, nothing is requested)
                if (c_item < item_end) { c_buf = c_buf == F_NSTAGE - 1 ? 0 : c_buf + 1; if (++c_k == nk) { c_k = 0; c_item += item_step; c_parity ^= 1; } }
            } else if (!(ABL & 4) && c_item < item_end) n_last = stage();
            const bool spread = (ABL & 4) && c_item < item_end;      // requests spread over the second MFMA half: one behind each of its first six MFMAs
            if (spread) stage_prep();
            if (kt + 1 < nk && !(ABL & 2)) rd(0, nxt);             // (the next TILE's first fragments are read after the epilogue: they would be 24 live registers in it)
            if constexpr ((ABL & 4) != 0) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
                        if (2 * i + j < 6 && spread) stage_issue(2 * i + j);
                    }
                __builtin_amdgcn_s_setprio(0);
                if (spread) n_last = stage_finish();
            } else {
                mma(1);
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        const bool inner = (cm0 + FBM <= p.M) && (cn0 + FBN <= p.N);
        const float* lbias = (const float*)(lds + F_BIAS_OFF + tile_parity * 1024) + wc * 64;
        auto run = [&](auto guard_tag) {
            constexpr bool G = decltype(guard_tag)::value;
            constexpr bool AUX_IN = (EPI == EPI_DQGELU_BF16);     // saved pre-activations of all eight 32x32 tiles requested up front
            uint4 auxr[AUX_IN ? 4 : 1][2][2];
            if constexpr (AUX_IN) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) epi_aux_load<G>(p, cm0 + i * 32, cn0 + wc * 64 + j * 32, lane, auxr[i][j]);
            }
            constexpr bool BIAS_PRE = (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_F32 || EPI == EPI_ACC_F32);
            f32x4 bq[2][4];
            if constexpr (BIAS_PRE && !LINES) { if (has_bias) epi_bias_preload(lbias, hi, bq); }
            if constexpr (LINES) {
                f32x4 bl[8];
                if (has_bias) epi_lines_bias_preload(lbias, hi, bl);
#pragma unroll
                for (int i = 0; i < 4; i++) epi_lines_bf16<EPI, G>(p, acc[i][0], acc[i][1], cm0 + i * 32, cn0 + wc * 64, lane, lbias, bl);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int64_t mt = cm0 + i * 32, nt = cn0 + wc * 64 + j * 32;
                        if constexpr (EPI == EPI_F32 || EPI == EPI_ACC_F32) {
                            epi_tile_f32<EPI, G>(p, acc[i][j], mt, nt, lane, lbias + j * 32, BIAS_PRE ? bq[j] : nullptr);
                        } else {
                            uint4 c0, c1;
                            epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32, AUX_IN ? auxr[AUX_IN ? i : 0][j] : nullptr, BIAS_PRE ? bq[j] : nullptr);
                            epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                            epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
                        }
                    }
                }
            }
        };
        if (inner) run(std::false_type{}); else run(std::true_type{});
        // known store count and no loads: the next two counted waits may leave the 16 stores in flight; anything else is followed by stricter waits
        // (a smaller count waits for everything older, stores included)
        store_age = (PLAIN_EPI && inner && !p.aux) ? 2 : 0;
        if (!(PLAIN_EPI && inner && !p.aux)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (loads / unknown store counts: keep the accounting exact)
        item += item_step;
        if (item >= item_end) break;
        tile_parity ^= 1;
    }
}

static int g_fr_abl = 0;
OWL_API int owl_gemm_fr_ablate(int a) { g_fr_abl = a; return 0; }
static int g_fr_slots = 512;             // persistent grid size
OWL_API int owl_gemm_fr_slots(int n) { g_fr_slots = n; return 0; }
static int g_fr_bw[2] = {0, 0};
OWL_API int owl_gemm_fr_block_width(int epi, int bw) { if (epi < 0 || epi > 1) return -1; g_fr_bw[epi] = bw; return 0; }

template <int EPI, bool LINES, int ABL>
static int launch_fr_abl(hipStream_t s, const GemmP& p, int nitems) {
    (void)hipFuncSetAttribute((const void*)gemm_fr_kernel<EPI, LINES, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    hipLaunchKernelGGL((gemm_fr_kernel<EPI, LINES, ABL>), dim3(p.persistent ? g_fr_slots : nitems), dim3(256), F_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

template <int EPI, bool LINES>
static int launch_fr_k(hipStream_t s, const GemmP& p, int nitems) {
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16) {
        switch (g_fr_abl) {
            case 1: return launch_fr_abl<EPI, LINES, 1>(s, p, nitems);
            case 2: return launch_fr_abl<EPI, LINES, 2>(s, p, nitems);
            case 3: return launch_fr_abl<EPI, LINES, 3>(s, p, nitems);
            case 4: return launch_fr_abl<EPI, LINES, 4>(s, p, nitems);
            case 8: return launch_fr_abl<EPI, LINES, 8>(s, p, nitems);
            case 12: return launch_fr_abl<EPI, LINES, 12>(s, p, nitems);
            case 16: return launch_fr_abl<EPI, LINES, 16>(s, p, nitems);
            case 20: return launch_fr_abl<EPI, LINES, 20>(s, p, nitems);
            default: break;
        }
    }
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_fr_kernel<EPI, LINES>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    });
    hipLaunchKernelGGL((gemm_fr_kernel<EPI, LINES>), dim3(p.persistent ? g_fr_slots : nitems), dim3(256), F_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
static int launch_fr(hipStream_t s, GemmP p) {
    p.tiles_m = (int)((p.M + FBM - 1) / FBM); p.tiles_n = (int)((p.N + FBN - 1) / FBN);
    p.dbg = 0;
    // column-block width of the tile order: as gemm_pp2.hip (blocks of up to 4 column tiles for the forward epilogues, row-major elsewhere)
    p.nsplit = p.tiles_n;
    if (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16)
        for (int d = 4; d >= 1; d--)
            if (p.tiles_n % d == 0) { p.nsplit = d; break; }
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16) {
        const int want = g_fr_bw[EPI == EPI_QGELU_BF16 ? 1 : 0];
        if (want > 0)
            for (int d = want; d >= 1; d--)
                if (p.tiles_n % d == 0) { p.nsplit = d; break; }
    }
    const int nitems = p.tiles_m * p.tiles_n;
    p.persistent = nitems > g_fr_slots ? 1 : 0;
    if constexpr (EPI == EPI_BIAS_BF16) {
        if (p.N % 8 == 0) return launch_fr_k<EPI, true>(s, p, nitems);
    }
    return launch_fr_k<EPI, false>(s, p, nitems);
}

// called from gemm.hip's dispatcher; returns 1 if this variant does not handle `epi` (or the shape: K must hold at least three K-steps of 32)
int owl_gemm_fr_launch(hipStream_t s, int epi, const GemmP& p) {
    if (p.K < 3 * FBK || p.K % FBK != 0) return 1;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_fr<EPI_BIAS_BF16>(s, p);
        case EPI_QGELU_BF16: return launch_fr<EPI_QGELU_BF16>(s, p);
        case EPI_DQGELU_BF16: return launch_fr<EPI_DQGELU_BF16>(s, p);
        case EPI_GELU_BF16: return launch_fr<EPI_GELU_BF16>(s, p);
        case EPI_DGELU_BF16: return launch_fr<EPI_DGELU_BF16>(s, p);
        case EPI_F32: return launch_fr<EPI_F32>(s, p);
        case EPI_ACC_F32: return launch_fr<EPI_ACC_F32>(s, p);
        default: return 1;
    }
}
#endif  // OWL_TUNING
