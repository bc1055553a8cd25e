// bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous; torch Linear layout)
// with fused epilogues.  One kernel template serves every dense contraction on the OWL-ViT train
// path (reference call sites: HF5:437-439,457 q/k/v/out proj; HF5:472,474 fc1/fc2; HF5:282-288
// patch-embed conv as an im2row-free GEMM; HF5:994-998 box head; ref src/models.py:25 class dense0;
// and their dX / dW backward forms).
//
// CDNA4 structure (template <EPI, BM, BN, WM, WN>):
//   * block tile BM x BN x 64, (BM/WM) x (BN/WN) waves, each wave a WM x WN sub-tile of
//     v_mfma_f32_32x32x16_bf16 accumulators.  Two instantiations: 256x256 with 8 waves of 128x64
//     (the workhorse: one K-step is 32 MFMAs per wave, long enough to cover the next tile's load
//     latency, and 25 % less LDS traffic per FLOP) and 128x128 with 4 waves of 64x64 (small / thin shapes);
//   * operand tiles go HBM -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction),
//     double-buffered, ONE barrier per K-step (preceded by vmcnt(0) AND lgkmcnt(0): LDS reads must have returned,
//     not just issued, before another wave may DMA into that buffer); the next tile's DMA is issued before the
//     current tile's MFMAs;
//   * LDS image of a [rows][64 k] bf16 tile is row-linear (the DMA destination is lane-linear) with the
//     16-byte chunk index XOR-swizzled by ((row>>1)&7) -- applied to the per-lane SOURCE address and again on
//     the ds_read_b128 address -- which makes every ds_read_b128 lane group conflict-free;
//   * blocks are remapped so that consecutive tiles of one A row-panel run on the same XCD (private L2).
#include "gemm_common.h"
#include <type_traits>

static constexpr int BK = 64;

// The shipped library has NO process-global mutable state (SURVEY.md section 8b "re-entrant"): the kernel choice is a per-call
// argument (`tile`) and the tuning switches below only exist in an OWL_TUNING build (csrc/build.sh with OWL_TUNING=1;
// prototypes in include/owl_hip_tuning.h), where tools/ uses them for A/B experiments.
#ifdef OWL_TUNING
static int g_persistent = 1;      // persistent scheduling (one workgroup per CU walks tiles with cross-tile prefetch)
OWL_API int owl_gemm_set_persistent(int on) { g_persistent = on; return 0; }
static int g_debug_slots = 0;     // override the persistent grid size
OWL_API int owl_gemm_debug_slots(int n) { g_debug_slots = n; return 0; }
static int g_debug_nostore = 0;   // 1: run the main loop but skip every epilogue store; 8: ping-pong trace run
OWL_API int owl_gemm_debug_nostore(int on) { g_debug_nostore = on; return 0; }
#else
static constexpr int g_persistent = 1, g_debug_slots = 0, g_debug_nostore = 0;
#endif

template <int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_nt_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool TRANS = (EPI == EPI_TRANS_BF16);
    constexpr bool PATCH = (EPI == EPI_PATCH_F32);
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN;
    constexpr int TI = WM / 32, TJ = WN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_Q = BM / 8 / NW, B_Q = BN / 8 / NW;   // LDS-DMA instructions per wave per stage

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int nk_all = (int)(p.K / BK);

    // ---- work items = (tile, K-split); persistent: this workgroup walks items item, item+step, ... of
    //      its XCD's contiguous chunk, so the 32 CUs of an XCD always sit on 32 consecutive tiles --------
    const int nitems = p.tiles_m * p.tiles_n * p.nsplit;
    int item, item_end, item_step;
    if (p.persistent) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, chunk = (nitems + 7) >> 3;
        item = xcd * chunk + idx;
        item_end = min(nitems, (xcd + 1) * chunk);
        item_step = gridDim.x >> 3;
    } else {
        item = xcd_remap(blockIdx.x, nitems);
        item_end = item + 1;
        item_step = 1;
    }
    if (item >= item_end) return;

    int64_t m0, n0;
    int split, kt0, kt1;
    auto decode = [&](int it) {
        const int tile = it / p.nsplit;
        split = it - tile * p.nsplit;
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
        m0 = (int64_t)tm * BM; n0 = (int64_t)tn * BN;
        kt0 = split * p.kt_per_split;
        kt1 = min(nk_all, kt0 + p.kt_per_split);
    };

    // ---- per-lane staging sources (row / swizzled chunk are K-step invariant) -----------------------
    const bf16_t* a_src[A_Q];
    const bf16_t* w_src[B_Q];
    auto setup_src = [&]() {
#pragma unroll
        for (int q = 0; q < A_Q; q++) {
            const int r = (w * A_Q + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int64_t am = m0 + r;
            if (am >= p.a_rows) am = p.a_rows - 1;
            if constexpr (PATCH) {
                const int64_t b = am / p.P, pp = am - b * p.P;
                const int64_t py = pp / p.G, px = pp - py * p.G;
                a_src[q] = p.A + ((b * 3) * p.S + py * p.ps) * p.S + px * p.ps;   // + (ch*S + ky)*S + kx per K-step
            } else {
                a_src[q] = p.A + am * p.lda + c * 8;
            }
        }
#pragma unroll
        for (int q = 0; q < B_Q; q++) {
            const int r = (w * B_Q + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int64_t wn = n0 + r;
            if (wn >= p.w_rows) wn = p.w_rows - 1;
            w_src[q] = p.W + wn * p.ldw + c * 8;
        }
    };

    // bias slice of a tile (BN floats) rides along with the tile's FIRST stage into LDS (parity-double-buffered),
    // so the epilogue reads it with ds_read and has no global load to wait for behind the prefetch DMA.
    auto stage_bias = [&](int parity) {
        if (p.bias && w == 0 && lane * 4 < BN) {
            int64_t n = n0 + lane * 4;
            if (n + 4 > p.N) n = p.N - 4;
            __builtin_amdgcn_global_load_lds(GPTR(p.bias + n), LPTR(lds + 2 * STAGE + parity * (BN * 4)), 16, 0, 0);
        }
    };
    auto stage = [&](int buf, int kt) {
        unsigned char* base = lds + buf * STAGE;
#pragma unroll
        for (int q = 0; q < A_Q; q++) {
            const int r0 = (w * A_Q + q) * 8;
            const bf16_t* ga;
            if constexpr (PATCH) {
                // k = kt*64 + c*8 -> (channel, ky, kx); an 8-pixel chunk never crosses a patch row (ps = 2^n >= 8)
                const int r = r0 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int k = kt * BK + c * 8;
                const int ch = k >> (2 * p.ps_log2), rem = k & ((1 << (2 * p.ps_log2)) - 1);
                const int ky = rem >> p.ps_log2, kx = rem & ((1 << p.ps_log2) - 1);
                ga = a_src[q] + ((int64_t)ch * p.S + ky) * p.S + kx;
            } else {
                ga = a_src[q] + (int64_t)kt * BK;
            }
            __builtin_amdgcn_global_load_lds(GPTR(ga), LPTR(base + r0 * 128), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_Q; q++) {
            const int r0 = (w * B_Q + q) * 8;
            __builtin_amdgcn_global_load_lds(GPTR(w_src[q] + (int64_t)kt * BK), LPTR(base + A_BYTES + r0 * 128), 16, 0, 0);
        }
    };

    const int wr = w / NWN, wc = w - wr * NWN;
    const int wr_ = wr, wc_ = wc;
    int a_off[TI], a_sw[TI], b_off[TJ], b_sw[TJ];
#pragma unroll
    for (int i = 0; i < TI; i++) {
        const int ra = wr * WM + i * 32 + (lane & 31);
        a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
        const int rb = wc * WN + j * 32 + (lane & 31);
        b_off[j] = A_BYTES + rb * 128; b_sw[j] = (rb >> 1) & 7;
    }

    // bf16-output epilogues on the 256-wide tile use the register path: element-wise math, pack, v_permlane32_swap
    // pairing -> 16-byte stores.  On gfx950 vmcnt counts stores too and is in-order, so the vmcnt(0) of the next K-step
    // drains this tile's 16 stores per wave before any new LDS-DMA can be consumed -- that drain (not the store issue
    // itself) is the epilogue cost of this single-phase kernel; gemm_pp.hip removes most of it with counted waits.
    // Holding the packed outputs in registers to store them during the next tile was measured and rejected (spills).
    constexpr bool WIDE = (BM == 256) && (EPI == EPI_BIAS_BF16 || EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16 ||
                                          EPI == EPI_DQGELU_BF16 || EPI == EPI_DGELU_BF16 || EPI == EPI_TRANS_BF16);

    decode(item);
    setup_src();
    stage(0, kt0);
    int tile_parity = 0;
    stage_bias(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;
    while (true) {
        f32x16 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        const int64_t cm0 = m0, cn0 = n0;
        const int csplit = split, ckt0 = kt0, ckt1 = kt1;
        const int next = item + item_step;
        const bool has_next = next < item_end;
        for (int kt = ckt0; kt < ckt1; kt++) {
            const bool last = (kt + 1 == ckt1);
            if (!last) {
                stage(cur ^ 1, kt + 1);
            } else if (has_next) {
                // cross-tile prefetch: the next item's first K-tile flies while this item finishes + stores
                decode(next);
                setup_src();
                stage(cur ^ 1, kt0);
                stage_bias(tile_parity ^ 1);
            }
            const unsigned char* tb = lds + cur * STAGE;
            // software-pipelined fragments: the six ds_read_b128 of K-chunk kc+1 are issued BEFORE the eight MFMAs of
            // chunk kc (sched_barrier pins the order; the compiler's own lgkmcnt(N) then waits only for the older set),
            // so a wave covers its LDS latency with its own matrix work instead of stalling every two MFMAs.
            bf16x8 fa[2][TI], fb[2][TJ];
            auto rd = [&](int set, int kc) {
                const int ch = kc * 2 + hi;
#pragma unroll
                for (int j = 0; j < TJ; j++) fb[set][j] = *(const bf16x8*)(tb + b_off[j] + ((ch ^ b_sw[j]) << 4));
#pragma unroll
                for (int i = 0; i < TI; i++) fa[set][i] = *(const bf16x8*)(tb + a_off[i] + ((ch ^ a_sw[i]) << 4));
            };
            rd(0, 0);
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const int cs = kc & 1;
                if (kc < 3) rd(cs ^ 1, kc + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TI; i++)
#pragma unroll
                    for (int j = 0; j < TJ; j++) {
                        if constexpr (TRANS)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cs][i], fb[cs][j], acc[i][j], 0, 0, 0);
                        else  // swapped: D rows = n, cols = m -> each lane owns 4 consecutive n of one m
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cs][j], fa[cs][i], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        if constexpr (WIDE) {
            const bool inner = (cm0 + BM <= p.M) && (cn0 + BN <= p.N);   // wave-uniform: no per-lane guards needed
            const float* lbias = (const float*)(lds + 2 * STAGE + tile_parity * (BN * 4)) + wc * WN;
            auto run = [&](auto guard_tag) {
                constexpr bool G = decltype(guard_tag)::value;
#pragma unroll
                for (int i = 0; i < TI; i++)
#pragma unroll
                    for (int j = 0; j < TJ; j++) {
                        const int t = i * TJ + j;
                        const int64_t mt = cm0 + wr * WM + i * 32, nt = cn0 + wc * WN + j * 32;
                        uint4 c0, c1;
                        epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32);
                        epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                        epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
                    }
            };
            if (inner) run(std::false_type{}); else run(std::true_type{});
        } else {
        // ---- epilogue AFTER the last barrier: buffer cur^1 was just consumed, and each wave stages through
        //      the slice of it that only IT will DMA into next; the global stores then drain under the next
        //      tile's first K-step (whose DMA is already in LDS buffer `cur`).
            unsigned char* xb = lds + (cur ^ 1) * STAGE;
            unsigned char* pieceA = xb + (w * A_Q * 8) * 128;
            unsigned char* pieceB = xb + A_BYTES + (w * B_Q * 8) * 128;
            static_assert(A_Q * 8 * 128 == 4096 && B_Q * 8 * 128 == 4096, "per-wave DMA slices must be 4 KiB each");
            static_assert(TJ == 2 && TI % 2 == 0, "a pass is two 32x32 tiles");
            if constexpr (TRANS) {
#pragma unroll
                for (int j = 0; j < TJ; j++)
#pragma unroll
                    for (int ip = 0; ip < TI / 2; ip++)
                        epi_pass<EPI>(p, acc[2 * ip][j], acc[2 * ip + 1][j], pieceA, pieceB, cn0 + wc * WN + j * 32,
                                      cm0 + wr * WM + ip * 64, lane, csplit);
            } else {
#pragma unroll
                for (int i = 0; i < TI; i++)
                    epi_pass<EPI>(p, acc[i][0], acc[i][1], pieceA, pieceB, cm0 + wr * WM + i * 32, cn0 + wc * WN, lane, csplit);
            }
        }
        if (!has_next) break;
        item = next;
        tile_parity ^= 1;
    }
}

template <int EPI, int BM, int BN, int WM, int WN>
static int launch_cfg(hipStream_t s, GemmP p, int splits) {
    constexpr int threads = (BM / WM) * (BN / WN) * 64;
    constexpr int lds_bytes = 2 * (BM + BN) * BK * 2 + 2 * BN * 4;   // 2 stages + 2 bias slices
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    p.tiles_m = (int)((p.M + BM - 1) / BM); p.tiles_n = (int)((p.N + BN - 1) / BN);
    p.nsplit = splits;
    if (g_debug_nostore == 1) p.M = 0;
    p.dbg = g_debug_nostore & ~1;            // tuning experiments: bit 1 = cache-resident store window, bit 2 = non-temporal bf16 stores
    const int nitems = p.tiles_m * p.tiles_n * splits;
    // persistent launch: one workgroup per CU-slot (blocks per CU limited by LDS), a multiple of 8 XCDs
    const int slots = g_debug_slots ? g_debug_slots : 256 * (lds_bytes > 80 * 1024 ? 1 : 2);
    p.persistent = (g_persistent && nitems > slots) ? 1 : 0;
    dim3 grid(p.persistent ? slots : nitems);
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, BM, BN, WM, WN>), grid, dim3(threads), lds_bytes, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

int owl_gemm_pp_launch(hipStream_t s, int epi, const GemmP& p, int slots_override, int persistent_on, int nostore);   // gemm_pp.hip
int owl_gemm_w4_launch(hipStream_t s, int epi, const GemmP& p);                                                       // gemm_w4.hip
int owl_gemm_pph_launch(hipStream_t s, int epi, const GemmP& p);                                                      // gemm_pph.hip
int owl_gemm_pp2_launch(hipStream_t s, int epi, const GemmP& p);                                                      // gemm_pp2.hip
int owl_gemm_fr_launch(hipStream_t s, int epi, const GemmP& p);                                                       // gemm_fr.hip

template <int EPI>
static int launch(hipStream_t s, const GemmP& p, int splits, int g_force_tile = 0) {
    // 256-wide tiles only when they give the chip enough work items (batch-1 out-proj is 10 x 3 of them: the 128x128
    // kernel's 114 tiles finish sooner)
    const int64_t t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const bool big = g_force_tile ? (g_force_tile == 256 || g_force_tile == 8 || g_force_tile == 9) : (p.M >= 512 && p.N >= 256 && t256 >= 48);
    if (big) return launch_cfg<EPI, 256, 256, 128, 64>(s, p, splits);
    return launch_cfg<EPI, 128, 128, 64, 64>(s, p, splits);
}

OWL_API int owl_gemm_nt_bf16(void* stream, int epi, const void* A, int64_t lda, int64_t a_rows, const void* W,
                                int64_t ldw, int64_t w_rows, const float* bias, void* out, int64_t ldo,
                                const float* resid, void* aux, int64_t ld_aux, int64_t M, int64_t N, int64_t K,
                                float alpha, int splits, int64_t Tp, int tile) {
    OWL_CHECK_ARG(A && W && out, "owl_gemm_nt_bf16: null pointer");
#ifdef OWL_TUNING
    OWL_CHECK_ARG(tile == 0 || tile == 128 || tile == 256 || tile == 8 || tile == 9 || tile == 4 || tile == 7 || tile == 5 || tile == 6, "owl_gemm_nt_bf16: tile must be 0 (auto), 6, 128, 256, 7 (or, tuning builds, 8, 9, 5, 4)");
#else
    OWL_CHECK_ARG(tile == 0 || tile == 6 || tile == 128 || tile == 256 || tile == 7, "owl_gemm_nt_bf16: tile must be 0 (auto), 6, 128, 256 or 7 (8, 9, 5, 4: the four-phase ping-pong, free-running and four-wave "
                                                                      "experiments exist only in an OWL_TUNING build)");
    OWL_CHECK_ARG(epi != EPI_TRANS_BF16 && epi != EPI_ATOMIC_F32, "owl_gemm_nt_bf16: epilogues 5 (f32 atomics) and 6 (per-head transposed) exist only in an OWL_TUNING build "
                                                                    "(the train path uses split-K slabs and reads V row-major)");
#endif
    const bool want_half = tile == 6;          // 6 = automatic + "half-height tiles if the whole problem is at most half a round" (see below)
    const int g_force_tile = want_half ? 0 : tile;
    OWL_CHECK_ARG(K > 0 && K % BK == 0, "owl_gemm_nt_bf16: K=%lld must be a positive multiple of 64", (long long)K);
    OWL_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0, "owl_gemm_nt_bf16: bad M=%lld N=%lld (N %% 8 == 0)", (long long)M, (long long)N);
    OWL_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "owl_gemm_nt_bf16: lda/ldw must be multiples of 8 elements");
    OWL_CHECK_ARG(a_rows > 0 && w_rows > 0, "owl_gemm_nt_bf16: a_rows / w_rows");
    OWL_CHECK_ARG(splits >= 1, "owl_gemm_nt_bf16: splits");
    GemmP p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.a_rows = a_rows;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.w_rows = w_rows;
    p.bias = bias; p.out = out; p.ldo = ldo; p.resid = resid; p.aux = aux; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.Tp = Tp;
    p.slab_stride = M * ldo;
    const int nk = (int)(K / BK);
    if (splits > nk) splits = nk;
    p.kt_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
    hipStream_t s = (hipStream_t)stream;
    const bool split_ok = (epi == EPI_ATOMIC_F32 || epi == EPI_SLAB_F32);
    OWL_CHECK_ARG(splits == 1 || split_ok, "owl_gemm_nt_bf16: split-K needs the slab epilogue");
    // bf16-output epilogues on big problems run the ping-pong schedule (gemm_pp.hip): 13-26 % faster, bit-identical
#ifdef OWL_TUNING
    if (g_force_tile == 4 && K >= 128) {                 // experimental four-wave kernel (tuning builds only)
        const int rc = owl_gemm_w4_launch(s, epi, p);
        if (rc <= 0) return rc;
    }
#endif
#ifdef OWL_TUNING
    if (g_force_tile == 5 && splits == 1) {              // round-4 experiment (tuning builds only): free-running 128 x 256 workgroups, two per CU (gemm_fr.hip)
        const int rc = owl_gemm_fr_launch(s, epi, p);
        if (rc <= 0) return rc;
    }
#endif
    if (g_force_tile == 7 && K >= 128) {                 // two-phase ping-pong kernel on the whole problem (A/B; falls through for other epilogues)
        const int rc = owl_gemm_pp2_launch(s, epi, p);
        if (rc <= 0) return rc;
    }
    // the ping-pong kernel for this epilogue: the two-phase schedule (gemm_pp2.hip) where it exists (bias / quick-GELU: +3..9 % on the model's
    // shapes, bit-identical), the four-phase one (gemm_pp.hip) otherwise or when asked for by tile = 8 / 9
    auto pp_launch = [&](const GemmP& q) -> int {
        if (g_force_tile == 0 && g_debug_nostore == 0) {
            const int rc = owl_gemm_pp2_launch(s, epi, q);
            if (rc <= 0) return rc;
        }
#ifdef OWL_TUNING
        return owl_gemm_pp_launch(s, epi, q, g_debug_slots, g_persistent, g_debug_nostore);      // the four-phase kernel (tile 8 / 9, the transposing epilogue)
#else
        return 1;                    // not a two-phase epilogue: the single-phase kernel below
#endif
    };
    const bool pp_auto = (g_force_tile == 0 && M >= 512 && N >= 256 && ((M + 255) / 256) * ((N + 255) / 256) >= 48);
    if ((g_force_tile == 8 || g_force_tile == 9 || pp_auto) && K >= 128 && (epi != EPI_TRANS_BF16 || (Tp > 0 && Tp % 4 == 0 && N % 64 == 0)) &&
        ((epi != EPI_DQGELU_BF16 && epi != EPI_DGELU_BF16) || aux)) {
        // Tile quantisation: tm x tn tiles of 256 x 256 on 256 workgroups = full_rounds whole rounds + a remainder round that keeps
        // only `rem` CUs busy (N = 768: 867 tiles = 3.39 -> 4 rounds).  When the remainder fits one round of HALF-height tiles the
        // whole rounds go to the 256 x 256 ping-pong kernel and the remaining row tiles to the 128 x 256 variant (gemm_pph.hip):
        // one round of ~0.56 tile times instead of a whole one.  Same K order and epilogue: bit-identical.
        // Measured (tools/gemm_remainder_bench.py, same-process A/B at M = 73 984): +3.6 % fc2 (K = 3072), +2.5 % dX (K = 2304), +4.2 % box-head
        // dense (GELU epilogue), +0.8 % out-proj (K = 768); nothing for wide outputs (QKV N = 2304: -0.1 %, fc1: does not fit one round), where
        // the half-height tiles -- latency-bound, ~0.85 of a full tile's time, not 0.56 -- only just pay for the second launch.  Hence the
        // automatic rule: narrow outputs only (N <= 1024); tile = 9 forces the split wherever it fits, tile = 8 never splits.
        // tile = 6 -- small problems (the reference's own batch size of 1: QKV = 90 tiles, fc1 = 120 on 256 CUs): no more 256 x 256 tiles than HALF the CUs ->
        // every tile goes out as two half-height tiles (gemm_pph.hip), one partial round of ~0.85 tile times on twice the CUs.  Same K order and epilogue:
        // bit-identical.  Asked for by the CALLER, who knows what else is in flight: with two sub-batch streams the other stream's tiles already fill the idle
        // CUs and the half-height split loses (forward batch 8: -2.7 %; alone: +3.3 % / +4.5 % on the batch-1 train step / forward, profiles/r05_small_batch.md).
        if (want_half && a_rows >= M && 2 * ((M + 255) / 256) * ((N + 255) / 256) <= NUM_CUS) {
            const int rc = owl_gemm_pph_launch(s, epi, p);
            if (rc <= 0) return rc;      // 1 = epilogue not handled by the half-height kernel: the 256 x 256 kernel below
        }
        if ((g_force_tile == 9 || (g_force_tile == 0 && N <= 1024)) && epi != EPI_TRANS_BF16 && epi != EPI_F32 && epi != EPI_ACC_F32 && a_rows >= M) {
            const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256, items = tm * tn;
            const int64_t full_rounds = items / NUM_CUS;
            const int64_t tm_main = (full_rounds * NUM_CUS) / tn;
            const int64_t rem_tiles = (tm - tm_main) * tn;
            if (full_rounds >= 1 && tm_main >= 1 && rem_tiles > 0 && 2 * rem_tiles <= NUM_CUS && items - full_rounds * NUM_CUS > 0) {
                const int64_t M_main = tm_main * 256;
                GemmP pm = p;
                pm.M = M_main; pm.a_rows = M_main;
                int rc = pp_launch(pm);
                if (rc < 0) return rc;
                if (rc == 0) {
                    GemmP pr = p;
                    pr.A = p.A + M_main * lda; pr.a_rows = a_rows - M_main; pr.M = M - M_main;
                    pr.out = (void*)((bf16_t*)p.out + M_main * ldo);
                    if (p.aux) pr.aux = (void*)((bf16_t*)p.aux + M_main * ld_aux);
                    rc = owl_gemm_pph_launch(s, epi, pr);
                    if (rc <= 0) return rc;
                    // (epilogue not handled by the half-height kernel: finish the remainder rows with the 256 x 256 kernel)
                    return pp_launch(pr);
                }
            }
        }
        const int rc = pp_launch(p);
        if (rc <= 0) return rc;      // 1 = epilogue not handled there: fall through
    }
    switch (epi) {
        case EPI_BIAS_BF16: return launch<EPI_BIAS_BF16>(s, p, 1, g_force_tile);
        case EPI_QGELU_BF16: return launch<EPI_QGELU_BF16>(s, p, 1, g_force_tile);
        case EPI_GELU_BF16: return launch<EPI_GELU_BF16>(s, p, 1, g_force_tile);
        case EPI_RESID_F32: OWL_CHECK_ARG(resid, "EPI_RESID_F32 needs resid"); return launch<EPI_RESID_F32>(s, p, 1, g_force_tile);
        case EPI_ACC_F32: p.resid = (const float*)out; return launch<EPI_ACC_F32>(s, p, 1, g_force_tile);
        case EPI_F32: return launch<EPI_F32>(s, p, 1, g_force_tile);
#ifdef OWL_TUNING
        case EPI_ATOMIC_F32: OWL_CHECK_ARG(!bias, "atomic epilogue takes no bias"); return launch<EPI_ATOMIC_F32>(s, p, splits, g_force_tile);
#endif
        case EPI_SLAB_F32: OWL_CHECK_ARG(!bias, "slab epilogue takes no bias"); return launch<EPI_SLAB_F32>(s, p, splits, g_force_tile);
#ifdef OWL_TUNING
        case EPI_TRANS_BF16:
            OWL_CHECK_ARG(Tp > 0 && Tp % 4 == 0 && N % 64 == 0, "EPI_TRANS_BF16: Tp %% 4, N %% 64");
            return launch<EPI_TRANS_BF16>(s, p, 1, g_force_tile);
#endif
        case EPI_DQGELU_BF16: OWL_CHECK_ARG(aux, "EPI_DQGELU needs aux"); return launch<EPI_DQGELU_BF16>(s, p, 1, g_force_tile);
        case EPI_DGELU_BF16: OWL_CHECK_ARG(aux, "EPI_DGELU needs aux"); return launch<EPI_DGELU_BF16>(s, p, 1, g_force_tile);
        default: owl_set_error("owl_gemm_nt_bf16: unknown epilogue %d", epi); return -1;
    }
}

// number of split-K slabs the call above will actually write for (K, splits): callers size the slab buffer with it
OWL_API int owl_gemm_effective_splits(int64_t K, int splits) {
    const int nk = (int)(K / BK);
    if (splits > nk) splits = nk;
    if (splits < 1) splits = 1;
    const int per = (nk + splits - 1) / splits;
    return (nk + per - 1) / per;
}

OWL_API int owl_gemm_slab_workspace_bytes(int64_t M, int64_t ldo, int64_t K, int splits, int64_t* bytes) {
    OWL_CHECK_ARG(bytes && M > 0 && ldo > 0 && K >= BK, "owl_gemm_slab_workspace_bytes: bad arguments");
    *bytes = (int64_t)owl_gemm_effective_splits(K, splits) * M * ldo * (int64_t)sizeof(float);
    return 0;
}

// out[i] (+)= sum_s slab[s][i]  -- deterministic split-K reduction (accumulate = 1 adds into out)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* out, int64_t n, int64_t stride, int nsplit, int accumulate) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = accumulate ? *(const float4*)(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = slabs + i;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {                 // 8 independent 16-byte loads in flight per thread (fixed summation order)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *(const float4*)(src + (int64_t)(s + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; u++) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; s < nsplit; s++) {
        const float4 v = *(const float4*)(src + (int64_t)s * stride);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *(float4*)(out + i) = a;
}

// The same reduction for TALL problems -- many partial rows, few columns (box_final_bwd: 1152 workgroup partials x 3844 columns at batch 32): slab_reduce_kernel
// gives every column quad ONE thread that walks all partials (4 workgroups on the chip, 54 us for 17 MB -- latency, not bandwidth; profiles/r06_tail.md).  Here a
// workgroup takes 32 column quads x 8 lanes per quad: lane g sums partials g, g + 8, g + 16, ... (fixed order), the eight sums meet in LDS and are added in order
// g = 0 .. 7 (fixed): deterministic, 8x the loads in flight per column and 8x the workgroups.  (Another summation order than slab_reduce_kernel's: the two are
// never mixed for one output -- the choice depends on the shape alone.)
__global__ __launch_bounds__(256) void tall_reduce_kernel(const float* __restrict__ slabs, float* out, int64_t n, int64_t stride, int nsplit, int accumulate) {
    __shared__ float4 part[8][32];
    const int q = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = ((int64_t)blockIdx.x * 32 + q) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const float* src = slabs + i;
        int sidx = g;
        for (; sidx + 24 < nsplit; sidx += 32) {           // 4 independent 16-byte loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *(const float4*)(src + (int64_t)(sidx + 8 * u) * stride);
#pragma unroll
            for (int u = 0; u < 4; u++) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; sidx < nsplit; sidx += 8) {
            const float4 v = *(const float4*)(src + (int64_t)sidx * stride);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    part[g][q] = a;
    __syncthreads();
    if (g == 0 && i < n) {
        float4 r = accumulate ? *(const float4*)(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; k++) { const float4 v = part[k][q]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        *(float4*)(out + i) = r;
    }
}

int owl_slab_reduce_impl(hipStream_t s, const float* slabs, float* out, int64_t n, int64_t slab_stride, int nsplit, int accumulate) {
    if (nsplit >= 128 && n / 4 <= 8192) {                    // tall: see tall_reduce_kernel
        hipLaunchKernelGGL(tall_reduce_kernel, dim3((unsigned)((n / 4 + 31) / 32)), dim3(256), 0, s, slabs, out, n, slab_stride, nsplit, accumulate);
        OWL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, slabs, out, n, slab_stride, nsplit, accumulate);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_slab_reduce(void* stream, const float* slabs, float* out, int64_t n, int64_t slab_stride, int nsplit, int accumulate) {
    OWL_CHECK_ARG(slabs && out && n > 0 && n % 4 == 0 && nsplit >= 1, "owl_slab_reduce: bad args (n %% 4 == 0)");
    return owl_slab_reduce_impl((hipStream_t)stream, slabs, out, n, slab_stride, nsplit, accumulate);
}

// Explicit im2row in the gather's K order (below) -- only for the single-phase REFERENCE kernels (tile = 256 / 128) on patch sizes that are not 2^n:
// patches[b*P + p, (c*ps + ky)*psp + pos] = image[b, c, py*ps + ky, px*ps + min(8*(pos/8), ps - 8) + pos%8], zero-padded to Kg.
__global__ __launch_bounds__(256) void im2row_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ out, int64_t B, int S, int ps, int psp,
                                                     int G, int K, int Kg) {
    const int64_t row = blockIdx.x;                  // b*P + p
    const int64_t P = (int64_t)G * G;
    const int64_t b = row / P; const int pp = (int)(row - b * P);
    const int py = pp / G, px = pp - py * G;
    for (int k = threadIdx.x; k < Kg; k += 256) {
        bf16_t v = 0;
        if (k < K) {
            const int r = k / psp, pos = k - r * psp, c = r / ps, ky = r - c * ps;
            const int kx = min((pos >> 3) << 3, ps - 8) + (pos & 7);
            v = img[((b * 3 + c) * S + py * ps + ky) * (int64_t)S + px * ps + kx];
        }
        out[row * Kg + k] = v;
    }
}

// Patch-embed: X[b*Tp + 1 + p, :] = W_pe . vec(patch(b,p)) + pos[1+p]   (no bias; HF5:282-288,336-343), im2row-free: the A loader gathers 16-byte runs of
// patch rows straight from the image (gemm_pp2.hip, stage_A) for every patch size >= 8.
//   ps = 2^n       : w_pe = the conv weight [D, 3*ps*ps] as it lies.
//   other ps (14)  : the K index pads a patch row to psp = 2^n >= ps positions; position `pos` of a row holds pixel min(8*(pos/8), ps - 8) + pos%8 (the last
//                    16-byte chunk overlaps its predecessor instead of leaving the row) and w_pe [D, Kg], Kg = 3*ps*psp rounded up to 64, holds the conv weight
//                    at each pixel's FIRST position and zeros elsewhere (Python: weights.patch_weight_gather_layout).
// `scratch` (bf16 [B*P (row-padded to 128), Kg]) is needed only when a single-phase reference kernel (tile = 256 / 128, or a problem too small for the ping-pong
// kernel) meets a patch size that is not 2^n: it then receives an explicit im2row in the same K order (identical bits).
static bool patch_embed_takes_pp2(int64_t M, int64_t D, int64_t Kg, int tile) {
    return (tile == 7 || (tile == 0 && M >= 512 && D >= 256 && ((M + 255) / 256) * ((D + 255) / 256) >= 48)) && Kg >= 128;
}

// Bytes of the `scratch` argument of owl_patch_embed_bf16 for this problem and kernel choice: 0 when the chosen kernel gathers from the image itself (every 2^n
// patch size; any patch size on the ping-pong kernel), else the im2row matrix bf16 [B*P rounded up to 128 rows, Kg].  The ONE place the rule lives (ADVICE r05:
// the Python side used to restate the dispatcher's predicate).
OWL_API int owl_patch_embed_scratch_bytes(int64_t B, int64_t S, int64_t ps, int64_t D, int tile, int64_t* bytes) {
    OWL_CHECK_ARG(bytes && B > 0 && ps >= 8 && ps <= 64 && ps % 2 == 0 && S > 0 && S % ps == 0 && D > 0, "owl_patch_embed_scratch_bytes: bad arguments");
    const int64_t G = S / ps, P = G * G;
    int64_t psp = 8; while (psp < ps) psp *= 2;
    const int64_t Kg = (3 * ps * psp + BK - 1) / BK * BK;
    *bytes = (psp == ps || patch_embed_takes_pp2(B * P, D, Kg, tile)) ? 0 : ((B * P + 127) / 128 * 128) * Kg * 2;
    return 0;
}

OWL_API int owl_patch_embed_bf16(void* stream, const void* image_bf16, const void* w_pe, const float* pos,
                                    float* x_out, void* scratch, int64_t B, int64_t S, int64_t ps, int64_t D, int64_t Tp, int tile) {
    OWL_CHECK_ARG(image_bf16 && w_pe && pos && x_out, "owl_patch_embed_bf16: null pointer");
    // (even patch sizes only: a 16-byte LDS-DMA piece starts at byte 2 * (row * S + px * ps + kx), which is 4-byte aligned -- what global_load_lds_dwordx4
    //  needs -- only when ps, hence S, is even; tested: 14, 16, 24, 32)
    OWL_CHECK_ARG(S % ps == 0 && ps >= 8 && ps <= 64 && ps % 2 == 0, "owl_patch_embed_bf16: image side must be a multiple of the patch size, patch size even and 8 <= patch size <= 64");
    const int64_t G = S / ps, P = G * G;
    int64_t psp = 8; while (psp < ps) psp *= 2;
    const int64_t K = 3 * ps * psp, Kg = (K + BK - 1) / BK * BK;
    OWL_CHECK_ARG(D % 8 == 0 && Tp >= P + 1, "owl_patch_embed_bf16: D %% 8, Tp");
    OWL_CHECK_ARG(B * 3 * S * S * 2 < (1LL << 32) && B * P < (1LL << 31), "owl_patch_embed_bf16: image batch beyond 4 GiB (unsigned 32-bit gather byte offsets)");
    OWL_CHECK_ARG(tile == 0 || tile == 7 || tile == 256 || tile == 128, "owl_patch_embed_bf16: tile must be 0 (auto), 7, 256 or 128");
    const bool pow2 = psp == ps;
    GemmP p{};
    p.bias = nullptr; p.out = x_out; p.ldo = D; p.M = B * P; p.N = D; p.alpha = 1.f;
    p.Tp = Tp; p.P = P; p.G = G; p.ps = ps; p.S = S; p.pos = pos;
    p.W = (const bf16_t*)w_pe; p.w_rows = D; p.a_rows = B * P;
    p.ps_log2 = 0; while ((1LL << p.ps_log2) < psp) p.ps_log2++;
    p.ps_magic = (int)(65536 / ps + 1);
    p.A = (const bf16_t*)image_bf16; p.lda = 0; p.ldw = Kg; p.K = Kg;
    p.kt_per_split = (int)(Kg / BK);
    p.nsplit = 1;
    // big problems: the two-phase ping-pong kernel (same bits); tile = 256 / 128 pins the single-phase kernels, 7 the ping-pong one
    if (patch_embed_takes_pp2(p.M, D, Kg, tile)) return owl_gemm_pp2_launch((hipStream_t)stream, EPI_PATCH_F32, p);
    if (pow2) return launch<EPI_PATCH_F32>((hipStream_t)stream, p, 1, tile);           // (the single-phase kernels gather 2^n patch rows themselves)
    OWL_CHECK_ARG(scratch, "owl_patch_embed_bf16: patch size %lld on a single-phase kernel (tile %d, or a problem too small for the ping-pong kernel) needs the im2row scratch buffer",
                  (long long)ps, tile);
    hipLaunchKernelGGL(im2row_kernel, dim3((unsigned)(B * P)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)image_bf16, (bf16_t*)scratch,
                       B, (int)S, (int)ps, (int)psp, (int)G, (int)K, (int)Kg);
    OWL_LAUNCH_CHECK();
    p.A = (const bf16_t*)scratch; p.lda = Kg;
    return launch<EPI_PATCHM_F32>((hipStream_t)stream, p, 1, tile);
}
