// Half-height (128 x 256 x 64) ping-pong GEMM tile for the REMAINDER ROUND of the 256 x 256 ping-pong kernel (gemm_pp.hip).
//
// Why: the model's N = 768 GEMMs (out-proj, fc2, their dX forms, the box head) have 289 x 3 = 867 tiles of 256 x 256 for 256
// workgroups = 3.39 rounds, executed as 4: the last round keeps 99 of 256 CUs busy.  The dispatcher (gemm.hip) therefore gives the
// whole rounds to the 256 x 256 kernel and the remaining row tiles to THIS kernel, split into half-height tiles: 99 tiles become 198
// workgroups of 128 x 256, one round of about 0.56 tile times instead of a whole one (3.39 -> 3.56 instead of 4 rounds).
//
// Same operands, same LDS image (LDS-DMA, 16-byte chunk XOR swizzle), same MFMA shape and K order, same epilogue code as gemm_pp.hip
// -> bit-identical outputs (tests/test_determinism_gpu.py).  Schedule: 8 waves = two groups of four (group g = output rows g*64..+64,
// wave wc = 64-column slice; every SIMD hosts one wave of each group); a K-tile is TWO quadrant phases (64x32 of the wave's 64x64
// output: 8 MFMAs of 32x32x16), each split by barriers into a LOAD half and an MFMA half; group 1 runs one barrier behind group 0, so
// one group owns the matrix pipe while the other waits for LDS / issues DMA:
//
//   half-slot:      4c           4c+1          4c+2          4c+3
//   group 0:    LOAD q0(c)    MFMA q0(c)    LOAD q1(c)    MFMA q1(c)
//   group 1:    MFMA q1(c-1)  LOAD q0(c)    MFMA q0(c)    LOAD q1(c)
//
// q0 LOAD reads A (both 32-row tiles, 8 x ds_read_b128) and B(j0) (4); q1 LOAD reads B(j1) (4).  Staging:
//   * a group's A rows are read by that group only and staged by that group's own waves, and the q0 LOAD ends with a barrier -> the A
//     pieces of K-tile c+2 go into the buffer of K-tile c during q1 LOAD of K-tile c (2 pieces per wave);
//   * B has THREE stages: the stage of K-tile c-1 is free once both groups have finished q1 LOAD of K-tile c-1, i.e. when group 0
//     enters q0 LOAD of K-tile c -> the B pieces of K-tile c+2 are issued there (4 pieces per wave), a whole K-tile of flight time;
//   * counted wait at the end of q1 LOAD (both groups: group 0 reads K-tile c+1 one barrier after group 1's q1 LOAD): K-tile c+1 has
//     landed, the six pieces of K-tile c+2 stay in flight across the barrier.
// One tile per workgroup (the remainder round has fewer tiles than CUs), so no cross-tile streaming.
#include "gemm_common.h"
#include <type_traits>

static constexpr int HBM = 128, HBN = 256, HBK = 64;
static constexpr int H_A_BYTES = HBM * HBK * 2, H_B_BYTES = HBN * HBK * 2;   // 16 / 32 KiB per stage
// LDS: two A stages, THREE B stages (B is staged two K-tiles ahead: with two it could only be issued one K-tile ahead -- its rows are
// shared by both groups and free late -- and the counted wait then stalled on it: the half-height K-tile took as long as a full one)
static constexpr int H_A_OFF = 0, H_B_OFF = 2 * H_A_BYTES, H_BIAS_OFF = H_B_OFF + 3 * H_B_BYTES, H_LDS = H_BIAS_OFF + 1024;

template <int N> __device__ __forceinline__ void ph_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ph_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ph_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pph_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int grp = w >> 2, wc = w & 3;
    const int nk = (int)(p.K / HBK);                 // >= 2 (host checks)
    const int nitems = p.tiles_m * p.tiles_n;
    const int item = xcd_remap(blockIdx.x, nitems);
    if (item >= nitems) return;
    const int tm = item / p.tiles_n, tn = item - tm * p.tiles_n;
    const int64_t cm0 = (int64_t)tm * HBM, cn0 = (int64_t)tn * HBN;

    // ---- staging: per-lane byte offsets relative to the tile's base pointers --------------------------------------------------
    const bf16_t* a_base = p.A + cm0 * p.lda;
    const bf16_t* w_base = p.W + cn0 * p.ldw;
    unsigned a_voff[2], w_voff[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int r = (w * 2 + q) * 8 + (lane >> 3);                 // A row 0..127: waves 0-3 stage group 0's rows, waves 4-7 group 1's
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int64_t am = cm0 + r; if (am >= p.a_rows) am = p.a_rows - 1;
        a_voff[q] = (unsigned)(((am - cm0) * p.lda + c * 8) * 2);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int rb = h * 128 + (w * 2 + q) * 8 + (lane >> 3);   // B row 0..255
            const int cb = (lane & 7) ^ ((rb >> 1) & 7);
            int64_t wn = cn0 + rb; if (wn >= p.w_rows) wn = p.w_rows - 1;
            w_voff[h][q] = (unsigned)(((wn - cn0) * p.ldw + cb * 8) * 2);
        }
    }
    const bool has_bias = p.bias != nullptr;
    unsigned b_voff = 0;
    const float* b_base = nullptr;
    if (has_bias) {
        int64_t n = cn0 + lane * 4; if (n + 4 > p.N) n = p.N - 4;
        b_base = p.bias + cn0;
        b_voff = (unsigned)((n - cn0) * 4);
    }
    auto stage_A = [&](int kt, int buf) {             // 2 pieces
        unsigned char* base = lds + H_A_OFF + buf * H_A_BYTES;
        const unsigned char* g = (const unsigned char*)(a_base + (int64_t)kt * HBK);
#pragma unroll
        for (int q = 0; q < 2; q++)
            __builtin_amdgcn_global_load_lds(GPTR(g + a_voff[q]), LPTR(base + ((w * 2 + q) * 8) * 128), 16, 0, 0);
    };
    auto stage_B = [&](int kt, int buf) -> int {      // 4 pieces (+ the bias slice with K-tile 0)
        unsigned char* base = lds + H_B_OFF + buf * H_B_BYTES;
        const unsigned char* g = (const unsigned char*)(w_base + (int64_t)kt * HBK);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds(GPTR(g + w_voff[h][q]), LPTR(base + (h * 128 + (w * 2 + q) * 8) * 128), 16, 0, 0);
        if (kt == 0 && has_bias) {
            __builtin_amdgcn_global_load_lds(GPTR((const unsigned char*)b_base + b_voff), LPTR(lds + H_BIAS_OFF), 16, 0, 0);
            return 5;
        }
        return 4;
    };

    // fragment addresses: A rows grp*64 + t*32 + (lane&31), B rows wc*64 + j*32 + (lane&31); one swizzle per operand
    const int a_row0 = grp * 64 + (lane & 31), b_row0 = wc * 64 + (lane & 31);
    const int a_base_off = a_row0 * 128, b_base_off = b_row0 * 128;
    const int a_swz = (a_row0 >> 1) & 7, b_swz = (b_row0 >> 1) & 7;

    // prologue: A(0), B(0) [+bias], A(1), B(1); retire A(0) and B(0), K-tile 1 stays in flight
    stage_A(0, 0);
    stage_B(0, 0);
    stage_A(1, 1);
    stage_B(1, 1);
    ph_wait<6>();
    ph_bar();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    if (grp == 1) ph_bar();                           // create the one-barrier offset between the groups
    int cur = 0, bcur = 0;                            // A stage (kt & 1), B stage (kt % 3)
    for (int kt = 0; kt < nk; kt++) {
        const unsigned char* ta = lds + H_A_OFF + cur * H_A_BYTES;
        const unsigned char* tbb = lds + H_B_OFF + bcur * H_B_BYTES;
        const int bnext2 = bcur >= 1 ? bcur - 1 : 2;  // (kt + 2) % 3
        bf16x8 fa[2][4], fb[2][4];                    // [32-row tile][kc], [j][kc]
        auto ld_a = [&]() {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int kc = 0; kc < 4; kc++) fa[t][kc] = *(const bf16x8*)(ta + a_base_off + t * 4096 + (((kc * 2 + hi) ^ a_swz) << 4));
        };
        auto ld_b = [&](int j) {
#pragma unroll
            for (int kc = 0; kc < 4; kc++) fb[j][kc] = *(const bf16x8*)(tbb + b_base_off + j * 4096 + (((kc * 2 + hi) ^ b_swz) << 4));
        };
        auto mma = [&](int j) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kc = 0; kc < 4; kc++)
#pragma unroll
                for (int t = 0; t < 2; t++)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kc], fa[t][kc], acc[t][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // ---- q0: the B stage of K-tile kt-1 is free (both groups are past q1 LOAD of that K-tile): it takes K-tile kt+2 ----
        ld_b(0); ld_a();
        const bool more = kt + 2 < nk;
        if (more) stage_B(kt + 2, bnext2);
        ph_wait_lgkm(); ph_bar();
        mma(0);
        ph_bar();
        // ---- q1: this group's A rows of this buffer are free (q0 LOAD ended with a barrier) ----
        ld_b(1);
        if (more) stage_A(kt + 2, cur);
        // Counted wait HERE, not one half-slot later: group 0 reads K-tile kt+1 right after the barrier that ends group 1's q1 LOAD, so
        // every wave of BOTH groups must have retired its pieces of K-tile kt+1 before the barrier that ends its own q1 LOAD (the two
        // groups run one barrier apart: "one barrier more", cdna_hip_programming.md).  Only the newest A pieces stay in flight.
        if (more) ph_wait<6>(); else ph_wait<0>();     // K-tile kt+1 has landed; K-tile kt+2 (4 B + 2 A pieces) stays in flight
        ph_bar();
        mma(1);
        ph_bar();
        cur ^= 1;
        bcur = bcur == 2 ? 0 : bcur + 1;
    }
    if (grp == 0) ph_bar();                           // let group 1 finish its last MFMA half: epilogues run together

    const bool inner = (cm0 + HBM <= p.M) && (cn0 + HBN <= p.N);
    const float* lbias = (const float*)(lds + H_BIAS_OFF) + wc * 64;
    auto run = [&](auto guard_tag) {
        constexpr bool G = decltype(guard_tag)::value;
        f32x4 bq[2][4];                                // the wave's bias values, one LDS round trip (gemm_common.h, epi_bias_preload)
        if (p.bias) epi_bias_preload(lbias, lane >> 5, bq);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t mt = cm0 + grp * 64 + i * 32, nt = cn0 + wc * 64 + j * 32;
                uint4 c0, c1;
                epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32, nullptr, bq[j]);
                epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
            }
    };
    if (inner) run(std::false_type{}); else run(std::true_type{});
}

template <int EPI>
static int launch_pph(hipStream_t s, GemmP p) {
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_pph_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS);
    });
    p.tiles_m = (int)((p.M + HBM - 1) / HBM); p.tiles_n = (int)((p.N + HBN - 1) / HBN);
    p.nsplit = 1; p.persistent = 0; p.dbg = 0;
    const int nitems = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_pph_kernel<EPI>), dim3(nitems), dim3(512), H_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

// called from gemm.hip's dispatcher for the remainder rows; returns 1 if this variant does not handle `epi`
int owl_gemm_pph_launch(hipStream_t s, int epi, const GemmP& p) {
    switch (epi) {
        case EPI_BIAS_BF16: return launch_pph<EPI_BIAS_BF16>(s, p);
        case EPI_QGELU_BF16: return launch_pph<EPI_QGELU_BF16>(s, p);
        case EPI_DQGELU_BF16: return launch_pph<EPI_DQGELU_BF16>(s, p);
        case EPI_GELU_BF16: return launch_pph<EPI_GELU_BF16>(s, p);
        case EPI_DGELU_BF16: return launch_pph<EPI_DGELU_BF16>(s, p);
        default: return 1;
    }
}
