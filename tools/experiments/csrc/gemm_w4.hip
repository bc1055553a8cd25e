// Four-wave variant of the 256x256x64 bf16 GEMM for the bf16-output epilogues (EXPERIMENTAL: owl_gemm_set_tile(4)).
//
// tools/pp_trace.py shows what bounds the ping-pong kernel (gemm_pp.hip): a wave's LOAD work per K-tile -- 24
// ds_read_b128 with four waves reading at once + 8 LDS-DMA pieces + waits ~ 1180 cycles -- exceeds the 4 x 288 cycles of
// its partner's MFMA halves.  This kernel cuts the LDS reads per MFMA from 0.75 to 0.5: ONE wave per SIMD owns a 128x128
// output block (4 x 4 accumulator tiles = 256 registers of the 512 a lone wave may use), reads 4 + 4 fragments per 16
// MFMAs, and hides them by software pipelining inside the wave (fragments of K-step kc+1 are requested before the MFMAs
// of step kc) instead of by a partner wave.  One barrier per K-tile (four MFMAs into step 2: K-tile c+1 visible, buffer c
// free); the 16 LDS-DMA pieces per wave of K-tile c+2 are spread four per K-step over steps 2, 3 and the next steps 0, 1.
//
// Measured (tools/w4_bench.py, bit-identical to the ping-pong kernel): with the staging removed the loop runs at 1490 TF/s
// at 8192^3 (ping-pong 1262, hipBLASLt 1670) -- the read side is fixed; with all 16 pieces in one K-step 890, in two 1047,
// in three 1197, in four 1185: the L1/TA path moves one 1 KiB piece per ~16 cycles for the whole CU (half the K-tile), and a lone wave
// per SIMD stalls its own MFMA stream whenever its piece queues behind another wave's.  At the model's K = 768 shapes the
// epilogue (one wave per SIMD, no partner to overlap with) loses more than the loop gains: 830 / 925 / 755 / 1050 TF/s vs
// 910 / 1000 / 930 / 1096 for out-proj / QK / fc1 / fc2.  Staggering the pieces per wave (wave w issues after the (w+1)-th
// MFMA of every group of four, one scalar branch per MFMA) measured WORSE (1088 at 8192^3).  NOT the default; kept as the starting point for round 2
// (stagger the pieces per wave, a leaner epilogue, 3 LDS stages of BK = 32).
#ifdef OWL_TUNING   // experimental kernel: part of tuning builds only (tools/w4_bench.py); the shipped library does not carry it
#include "gemm_common.h"
#include <type_traits>

static constexpr int W4_A_BYTES = 256 * 64 * 2, W4_STAGE = 2 * W4_A_BYTES;            // 32 + 32 KiB
static constexpr int W4_BIAS_OFF = 2 * W4_STAGE, W4_LDS = 2 * W4_STAGE + 2 * 1024;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int wr = w >> 1, wc = w & 1;               // 128-row / 128-column block of the tile
    const int nk = (int)(p.K / 64);                  // >= 2 (host checks)
    const int nitems = p.tiles_m * p.tiles_n;
    int item, item_end, item_step;
    if (p.persistent) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, chunk = (nitems + 7) >> 3;
        item = xcd * chunk + idx; item_end = min(nitems, (xcd + 1) * chunk); item_step = gridDim.x >> 3;
    } else {
        item = xcd_remap(blockIdx.x, nitems); item_end = item + 1; item_step = 1;
    }
    if (item >= item_end) return;

    // ---- DMA stream: K-tiles in consumption order across the persistent tile loop, two K-tiles ahead ---------------
    // piece t (0..7) of an operand: rows t*32 + w*8 + (lane>>3); 16-byte chunks XOR-swizzled by ((row>>1)&7)
    int s_item = item, s_k = 0, s_buf = 0, s_parity = 0;
    unsigned a_voff[8], w_voff[8], b_voff = 0;
    __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t w_rsrc = a_rsrc, b_rsrc = a_rsrc;
    const bool has_bias = p.bias != nullptr;
    auto stream_setup = [&]() {
        const int tm = s_item / p.tiles_n, tn = s_item - tm * p.tiles_n;
        const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * p.lda), 0, 0x7fffffff, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
        const int a_last = (int)min((int64_t)255, p.a_rows - 1 - m0), w_last = (int)min((int64_t)255, p.w_rows - 1 - n0);
        const int lda2 = (int)p.lda * 2, ldw2 = (int)p.ldw * 2;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int r = t * 32 + w * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            a_voff[t] = (unsigned)(min(r, a_last) * lda2 + c * 16);
            w_voff[t] = (unsigned)(min(r, w_last) * ldw2 + c * 16);
        }
        if (has_bias) {
            const int nrem = (int)min((int64_t)256, p.N - n0);
            b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias + n0), 0, 0x7fffffff, 0x00020000);
            b_voff = (unsigned)(min(lane * 4, nrem - 4) * 4);
        }
    };
    auto stream_live = [&]() { return s_item < item_end; };
    auto stage_piece = [&](int t) {                  // t = 0..7 A pieces, 8..15 B pieces (compile-time under unrolling)
        if (t < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LPTR(lds + s_buf * W4_STAGE + (t * 32 + w * 8) * 128), 16, (int)a_voff[t], s_k * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LPTR(lds + s_buf * W4_STAGE + W4_A_BYTES + ((t - 8) * 32 + w * 8) * 128), 16,
                                                     (int)w_voff[t - 8], s_k * 128, 0, 0);
    };
    auto stage_begin = [&]() {                       // before the first piece of a K-tile
        if (s_k == 0) {
            stream_setup();
            if (has_bias) __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, LPTR(lds + W4_BIAS_OFF + s_parity * 1024), 16, (int)b_voff, 0, 0, 0);
        }
    };
    auto stage_end = [&]() {
        s_buf ^= 1;
        if (++s_k == nk) { s_k = 0; s_item += item_step; s_parity ^= 1; }
    };
    auto stage_all = [&]() {
        stage_begin();
#pragma unroll
        for (int t = 0; t < 16; t++) stage_piece(t);
        stage_end();
    };

    // fragment addresses: A rows wr*128 + i*32 + (lane&31), B rows wc*128 + j*32 + (lane&31); the swizzle term is the same for
    // every 32-row tile of a lane -> one address per K-step and operand, tiles by immediate offsets
    typedef const __attribute__((address_space(3))) bf16x8* frag_ptr;
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(lds);
    const int a_row0 = wr * 128 + (lane & 31), b_row0 = wc * 128 + (lane & 31);
    unsigned a_addr[4], b_addr[4];
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        a_addr[kc] = lds0 + a_row0 * 128 + (((kc * 2 + hi) ^ ((a_row0 >> 1) & 7)) << 4);
        b_addr[kc] = lds0 + W4_A_BYTES + b_row0 * 128 + (((kc * 2 + hi) ^ ((b_row0 >> 1) & 7)) << 4);
    }

    bf16x8 fa[2][4], fb[2][4];
    auto load_frags = [&](int slot, int buf, int kc) {
#pragma unroll
        for (int i = 0; i < 4; i++) fa[slot][i] = *(frag_ptr)(uintptr_t)(a_addr[kc] + buf * W4_STAGE + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; j++) fb[slot][j] = *(frag_ptr)(uintptr_t)(b_addr[kc] + buf * W4_STAGE + j * 4096);
    };

    // prologue: K-tiles 0 and 1 in flight, K-tile 0 landed, its first fragments requested
    stage_all();
    const bool two = stream_live();
    if (two) stage_all();
    if (two) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0, tile_parity = 0;
    bool pend = false;              // pieces 8-15 of the K-tile being staged are still to be issued (next steps 0 and 1)
    load_frags(0, 0, 0);

    while (true) {
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        const int tm = item / p.tiles_n, tn = item - tm * p.tiles_n;
        const int64_t cm0 = (int64_t)tm * 256, cn0 = (int64_t)tn * 256;
        const bool last_item = item + item_step >= item_end;
        for (int kt = 0; kt < nk; kt++) {
            const bool more = !(last_item && kt + 1 == nk);       // another K-tile follows in the stream
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const int slot = kc & 1;
                // request the next K-step's fragments first
                if (kc < 3) load_frags(slot ^ 1, cur, kc + 1);
                else if (more) load_frags(slot ^ 1, cur ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);          // keep the requests a whole K-step ahead of their use
                // K-tile c+2 goes into the buffer this K-tile is read from.  Its last reads (the step-3 fragments) are issued at the
                // top of step 2, so ONE barrier four MFMAs into step 2 both releases it (every wave's reads have returned) and
                // publishes K-tile c+1 (every wave's pieces have landed: they were issued 3-4 K-steps ago).  Pieces 0-7 (A) go out
                // between the remaining MFMAs of step 2, pieces 8-15 (B) between those of step 3.
                // issue plan (m = MFMA index of the step): step 2: pieces 0-3 after m = 6, 9, 12, 15; steps 3, 0', 1': four pieces
                // each after m = 1, 5, 9, 13.  The L1/TA path moves 1 KiB per ~16 cycles for the whole CU, i.e. the 64 pieces of a
                // K-tile occupy it for half the K-tile: bunching them stalls the issuing waves.
                const bool dma = (kc >= 2) && stream_live();
                const bool dma0 = (kc <= 1) && pend;
                if (kc == 2 && dma) stage_begin();
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int m = i * 4 + j;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][j], fa[slot][i], acc[i][j], 0, 0, 0);
                        if (kc == 2 && m == 3) {
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (kc == 2 && m >= 6 && (m % 3) == 0) { if (dma) stage_piece((m - 6) / 3); }
                        if (kc == 3 && (m & 3) == 1) { if (dma) stage_piece(4 + (m >> 2)); }
                        if (kc == 0 && (m & 3) == 1) { if (dma0) stage_piece(8 + (m >> 2)); }
                        if (kc == 1 && (m & 3) == 1) { if (dma0) stage_piece(12 + (m >> 2)); }
                    }
                if (kc == 3 && dma) pend = true;
                if (kc == 1 && dma0) { stage_end(); pend = false; }
                __builtin_amdgcn_sched_barrier(0);
            }
            cur ^= 1;
        }
        // ---- epilogue: 16 accumulator tiles, register-resident bf16 conversion, 16-byte stores --------------------------
        const bool inner = (cm0 + 256 <= p.M) && (cn0 + 256 <= p.N);
        const float* lbias = (const float*)(lds + W4_BIAS_OFF + tile_parity * 1024) + wc * 128;
        auto run = [&](auto guard_tag) {
            constexpr bool G = decltype(guard_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int64_t mt = cm0 + wr * 128 + i * 32, nt = cn0 + wc * 128 + j * 32;
                    uint4 c0, c1;
                    epi_tile_bf16<EPI, G>(p, acc[i][j], mt, nt, lane, c0, c1, lbias + j * 32);
                    epi_store_chunk<EPI, G>(p, c0, mt, nt, 0, lane);
                    epi_store_chunk<EPI, G>(p, c1, mt, nt, 1, lane);
                }
        };
        if (inner) run(std::false_type{}); else run(std::true_type{});
        item += item_step;
        if (item >= item_end) break;
        tile_parity ^= 1;
    }
}

template <int EPI>
static int launch_w4(hipStream_t s, GemmP p) {
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, {
        (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    });
    p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (int)((p.N + 255) / 256);
    p.nsplit = 1; p.dbg = 0;
    const int nitems = p.tiles_m * p.tiles_n;
    p.persistent = nitems > 256 ? 1 : 0;
    hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3(p.persistent ? 256 : nitems), dim3(256), W4_LDS, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

// called from gemm.hip's dispatcher; returns 1 if this variant does not handle `epi`
int owl_gemm_w4_launch(hipStream_t s, int epi, const GemmP& p) {
    switch (epi) {
        case EPI_BIAS_BF16: return launch_w4<EPI_BIAS_BF16>(s, p);
        case EPI_QGELU_BF16: return launch_w4<EPI_QGELU_BF16>(s, p);
        default: return 1;
    }
}

#endif  // OWL_TUNING
