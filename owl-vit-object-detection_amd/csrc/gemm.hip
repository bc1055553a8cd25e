// bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous; torch Linear layout)
// with fused epilogues.  One kernel template serves every dense contraction on the OWL-ViT train
// path (reference call sites: HF5:437-439,457 q/k/v/out proj; HF5:472,474 fc1/fc2; HF5:282-288
// patch-embed conv as an im2row-free GEMM; HF5:994-998 box head; ref src/models.py:25 class dense0;
// and their dX / dW backward forms).
//
// CDNA4 structure: 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16 accumulators.  Operand tiles go HBM -> LDS by direct LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction), double-buffered, one barrier per K-step.
// The LDS image of a [128 rows][64 k] bf16 tile is row-linear (the DMA destination is
// lane-linear) with the 16-byte chunk index XOR-swizzled by ((row>>1)&7) -- applied to the per-lane
// SOURCE address and again on the ds_read_b128 address -- which makes every ds_read_b128 lane group
// conflict-free.  Blocks are remapped so that consecutive tiles of one A row-panel run on the same
// XCD (private L2).
#include "common.h"

enum {
    EPI_BIAS_BF16 = 0,   // out bf16 = acc + bias
    EPI_QGELU_BF16 = 1,  // u = acc + bias; out bf16 = u*sigmoid(1.702u); aux (optional) bf16 = u
    EPI_GELU_BF16 = 2,   // erf GELU, aux (optional) bf16 = u
    EPI_RESID_F32 = 3,   // out f32 = resid + acc + bias
    EPI_F32 = 4,         // out f32 = alpha*acc (+ bias)
    EPI_ATOMIC_F32 = 5,  // atomicAdd(out f32, alpha*acc)            (split-K dW)
    EPI_TRANS_BF16 = 6,  // out_t[b][n/64][n%64][t] bf16 = acc + bias, m = b*Tp + t   (per-head transposed)
    EPI_PATCH_F32 = 7,   // A gathered from image patches; out f32 [b*Tp + 1 + p][n] = acc + pos[1+p][n]
    EPI_DQGELU_BF16 = 8, // out bf16 = acc * quick_gelu'(aux u)
    EPI_DGELU_BF16 = 9,  // out bf16 = acc * gelu_erf'(aux u)
    EPI_ACC_F32 = 10,    // out f32 += acc   (resid == out)
};

struct GemmP {
    const bf16_t* A; int64_t lda; int64_t a_rows;
    const bf16_t* W; int64_t ldw; int64_t w_rows;
    const float* bias;
    void* out; int64_t ldo;
    const float* resid;
    void* aux; int64_t ld_aux;
    int64_t M, N, K;       // M,N: store guards; K multiple of 64
    int tiles_m, tiles_n, kt_per_split;
    float alpha;
    // EPI_TRANS
    int64_t Tp;            // rows per image
    // EPI_PATCH
    int64_t P, G, ps, S;   // patches / grid / patch size / image side
    int ps_log2;
    const float* pos;      // [T, N]
};

static constexpr int BM = 128, BN = 128, BK = 64;
static constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB

__device__ __forceinline__ float qgelu_f(float u) { return u / (1.0f + __expf(-1.702f * u)); }
__device__ __forceinline__ float dqgelu_f(float u) {
    float s = 1.0f / (1.0f + __expf(-1.702f * u));
    return s * (1.0f + 1.702f * u * (1.0f - s));
}
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.39894228040143268f * __expf(-0.5f * u * u);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool TRANS = (EPI == EPI_TRANS_BF16);
    constexpr bool PATCH = (EPI == EPI_PATCH_F32);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- XCD-aware tile mapping (bijective; blocks b, b+8, ... share an XCD) -----------------
    const int ntile = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int nk_all = (int)(p.K / BK);
    const int kt0 = blockIdx.y * p.kt_per_split;
    const int kt1 = min(nk_all, kt0 + p.kt_per_split);
    if (kt0 >= kt1) return;

    // ---- per-lane staging sources (row / swizzled chunk are K-step invariant) -----------------
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = (w * 4 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int64_t am = m0 + r;
        if (am >= p.a_rows) am = p.a_rows - 1;
        if constexpr (PATCH) {
            // A row = patch (b, py, px); chunk c covers 8 pixels of one patch row (ps % 8 == 0)
            const int64_t b = am / p.P, pp = am - b * p.P;
            const int64_t py = pp / p.G, px = pp - py * p.G;
            // k-dependent part added per K-step; keep the (b, py, px) base here
            a_src[q] = p.A + ((b * 3) * p.S + py * p.ps) * p.S + px * p.ps;  // + (ch*S + ky)*S + kx per K-step
        } else {
            a_src[q] = p.A + am * p.lda + c * 8;
        }
        int64_t wn = n0 + r;
        if (wn >= p.w_rows) wn = p.w_rows - 1;
        w_src[q] = p.W + wn * p.ldw + c * 8;
    }

    auto stage = [&](int buf, int kt) {
        unsigned char* base = lds + buf * (2 * TILE_BYTES);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r0 = (w * 4 + q) * 8;
            const bf16_t* ga;
            if constexpr (PATCH) {
                // k = kt*64 + c*8 -> (channel, ky, kx); an 8-pixel chunk never crosses a patch row
                // because ps is a power of two >= 8
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
            const bf16_t* gw = w_src[q] + (int64_t)kt * BK;
            __builtin_amdgcn_global_load_lds(GPTR(gw), LPTR(base + TILE_BYTES + r0 * 128), 16, 0, 0);
        }
    };

    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes within a tile), K-chunk XOR applied per kc below
    int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int ra = wr * 64 + i * 32 + (lane & 31);
        const int rb = wc * 64 + i * 32 + (lane & 31);
        a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
        b_off[i] = rb * 128; b_sw[i] = (rb >> 1) & 7;
    }
    const int hi = lane >> 5;

    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;
    for (int kt = kt0; kt < kt1; kt++) {
        if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
        const unsigned char* ta = lds + cur * (2 * TILE_BYTES);
        const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
            const int ch = kc * 2 + hi;
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                af[i] = *(const bf16x8*)(ta + a_off[i] + ((ch ^ a_sw[i]) << 4));
                bfr[i] = *(const bf16x8*)(tb + b_off[i] + ((ch ^ b_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if constexpr (TRANS)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    else  // swapped: D rows = n, cols = m  -> each lane owns 4 consecutive n of one m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const float alpha = p.alpha;
    if constexpr (TRANS) {
        // D[row = m_local][col = n_local]; lane: n = lane&31, m quads
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t n = n0 + wc * 64 + j * 32 + (lane & 31);
                if (n >= p.N) continue;
                const float bv = p.bias ? p.bias[n] : 0.f;
                bf16_t* orow = (bf16_t*)p.out + n * p.Tp;   // + b * N * Tp + t
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const int64_t m = m0 + wr * 64 + i * 32 + 8 * qd + 4 * hi;
                    if (m >= p.M) continue;
                    const int64_t b = m / p.Tp, t = m - b * p.Tp;
                    uint2 v;
                    v.x = pack_bf2(acc[i][j][qd * 4 + 0] + bv, acc[i][j][qd * 4 + 1] + bv);
                    v.y = pack_bf2(acc[i][j][qd * 4 + 2] + bv, acc[i][j][qd * 4 + 3] + bv);
                    *(uint2*)(orow + b * p.N * p.Tp + t) = v;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t m = m0 + wr * 64 + i * 32 + (lane & 31);
            if (m >= p.M) continue;
            int64_t orow_idx = m;
            const float* posrow = nullptr;
            if constexpr (PATCH) {
                const int64_t b = m / p.P, pp = m - b * p.P;
                orow_idx = b * p.Tp + 1 + pp;
                posrow = p.pos + (1 + pp) * p.N;
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const int64_t n = n0 + wc * 64 + j * 32 + 8 * qd + 4 * hi;
                    if (n >= p.N) continue;   // N is a multiple of 4 (checked on the host)
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = acc[i][j][qd * 4 + e] * alpha;
                    if (p.bias) {
                        const float4 b4 = *(const float4*)(p.bias + n);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    if constexpr (EPI == EPI_BIAS_BF16) {
                        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        *(uint2*)((bf16_t*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_QGELU_BF16 || EPI == EPI_GELU_BF16) {
                        if (p.aux) {
                            uint2 a; a.x = pack_bf2(v[0], v[1]); a.y = pack_bf2(v[2], v[3]);
                            *(uint2*)((bf16_t*)p.aux + orow_idx * p.ld_aux + n) = a;
                        }
                        float g[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) g[e] = (EPI == EPI_QGELU_BF16) ? qgelu_f(v[e]) : gelu_f(v[e]);
                        uint2 o; o.x = pack_bf2(g[0], g[1]); o.y = pack_bf2(g[2], g[3]);
                        *(uint2*)((bf16_t*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_DQGELU_BF16 || EPI == EPI_DGELU_BF16) {
                        const uint2 a = *(const uint2*)((const bf16_t*)p.aux + orow_idx * p.ld_aux + n);
                        float u[4] = {bf2f(a.x & 0xffff), bf2f(a.x >> 16), bf2f(a.y & 0xffff), bf2f(a.y >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] *= (EPI == EPI_DQGELU_BF16) ? dqgelu_f(u[e]) : dgelu_f(u[e]);
                        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        *(uint2*)((bf16_t*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_ACC_F32) {
                        const float4 r4 = *(const float4*)(p.resid + orow_idx * p.ldo + n);
                        float4 o = {r4.x + v[0], r4.y + v[1], r4.z + v[2], r4.w + v[3]};
                        *(float4*)((float*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_F32) {
                        float4 o = {v[0], v[1], v[2], v[3]};
                        *(float4*)((float*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_PATCH_F32) {
                        const float4 p4 = *(const float4*)(posrow + n);
                        float4 o = {v[0] + p4.x, v[1] + p4.y, v[2] + p4.z, v[3] + p4.w};
                        *(float4*)((float*)p.out + orow_idx * p.ldo + n) = o;
                    } else if constexpr (EPI == EPI_ATOMIC_F32) {
                        float* o = (float*)p.out + orow_idx * p.ldo + n;
#pragma unroll
                        for (int e = 0; e < 4; e++) atomicAdd(o + e, v[e]);
                    }
                }
        }
    }
}

template <int EPI>
static int launch(hipStream_t s, const GemmP& p, int splits) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        attr_done = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, splits);
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, grid, dim3(256), 4 * TILE_BYTES, s, p);
    OWL_LAUNCH_CHECK();
    return 0;
}

extern "C" int owl_gemm_nt_bf16(void* stream, int epi, const void* A, int64_t lda, int64_t a_rows, const void* W,
                                int64_t ldw, int64_t w_rows, const float* bias, void* out, int64_t ldo,
                                const float* resid, void* aux, int64_t ld_aux, int64_t M, int64_t N, int64_t K,
                                float alpha, int splits, int64_t Tp) {
    OWL_CHECK_ARG(A && W && out, "owl_gemm_nt_bf16: null pointer");
    OWL_CHECK_ARG(K > 0 && K % BK == 0, "owl_gemm_nt_bf16: K=%lld must be a positive multiple of 64", (long long)K);
    OWL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0, "owl_gemm_nt_bf16: bad M=%lld N=%lld (N %% 4 == 0)", (long long)M, (long long)N);
    OWL_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "owl_gemm_nt_bf16: lda/ldw must be multiples of 8 elements");
    OWL_CHECK_ARG(a_rows > 0 && w_rows > 0, "owl_gemm_nt_bf16: a_rows / w_rows");
    OWL_CHECK_ARG(splits >= 1, "owl_gemm_nt_bf16: splits");
    GemmP p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.a_rows = a_rows;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.w_rows = w_rows;
    p.bias = bias; p.out = out; p.ldo = ldo; p.resid = resid; p.aux = aux; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.Tp = Tp;
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
    const int nk = (int)(K / BK);
    if (splits > nk) splits = nk;
    p.kt_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
    hipStream_t s = (hipStream_t)stream;
    switch (epi) {
        case EPI_BIAS_BF16: OWL_CHECK_ARG(splits == 1, "split-K needs the atomic epilogue"); return launch<EPI_BIAS_BF16>(s, p, 1);
        case EPI_QGELU_BF16: OWL_CHECK_ARG(splits == 1, "split-K needs the atomic epilogue"); return launch<EPI_QGELU_BF16>(s, p, 1);
        case EPI_GELU_BF16: OWL_CHECK_ARG(splits == 1, "split-K needs the atomic epilogue"); return launch<EPI_GELU_BF16>(s, p, 1);
        case EPI_RESID_F32:
            OWL_CHECK_ARG(splits == 1 && resid, "EPI_RESID_F32 needs resid, no split-K");
            return launch<EPI_RESID_F32>(s, p, 1);
        case EPI_ACC_F32:
            OWL_CHECK_ARG(splits == 1, "no split-K"); p.resid = (const float*)out;
            return launch<EPI_ACC_F32>(s, p, 1);
        case EPI_F32: OWL_CHECK_ARG(splits == 1, "split-K needs the atomic epilogue"); return launch<EPI_F32>(s, p, 1);
        case EPI_ATOMIC_F32:
            OWL_CHECK_ARG(!bias, "atomic epilogue takes no bias");
            return launch<EPI_ATOMIC_F32>(s, p, splits);
        case EPI_TRANS_BF16:
            OWL_CHECK_ARG(splits == 1 && Tp > 0 && Tp % 4 == 0 && N % 64 == 0, "EPI_TRANS_BF16: Tp %% 4, N %% 64");
            return launch<EPI_TRANS_BF16>(s, p, 1);
        case EPI_DQGELU_BF16:
            OWL_CHECK_ARG(splits == 1 && aux, "EPI_DQGELU needs aux"); return launch<EPI_DQGELU_BF16>(s, p, 1);
        case EPI_DGELU_BF16:
            OWL_CHECK_ARG(splits == 1 && aux, "EPI_DGELU needs aux"); return launch<EPI_DGELU_BF16>(s, p, 1);
        default: owl_set_error("owl_gemm_nt_bf16: unknown epilogue %d", epi); return -1;
    }
}

// Patch-embed: X[b*Tp + 1 + p, :] = W_pe . vec(patch(b,p)) + pos[1+p]   (no bias; HF5:282-288,336-343)
extern "C" int owl_patch_embed_bf16(void* stream, const void* image_bf16, const void* w_pe, const float* pos,
                                    float* x_out, int64_t B, int64_t S, int64_t ps, int64_t D, int64_t Tp) {
    OWL_CHECK_ARG(image_bf16 && w_pe && pos && x_out, "owl_patch_embed_bf16: null pointer");
    OWL_CHECK_ARG(ps >= 8 && (ps & (ps - 1)) == 0 && S % ps == 0,
                  "owl_patch_embed_bf16: fused loader needs a power-of-two patch size >= 8 (got %lld)", (long long)ps);
    const int64_t G = S / ps, P = G * G, K = 3 * ps * ps;
    OWL_CHECK_ARG(K % BK == 0 && D % 4 == 0 && Tp >= P + 1, "owl_patch_embed_bf16: K %% 64, D %% 4, Tp");
    GemmP p{};
    p.A = (const bf16_t*)image_bf16; p.lda = 0; p.a_rows = B * P;
    p.W = (const bf16_t*)w_pe; p.ldw = K; p.w_rows = D;
    p.bias = nullptr; p.out = x_out; p.ldo = D; p.M = B * P; p.N = D; p.K = K; p.alpha = 1.f;
    p.Tp = Tp; p.P = P; p.G = G; p.ps = ps; p.S = S; p.pos = pos;
    p.ps_log2 = 0; while ((1LL << p.ps_log2) < ps) p.ps_log2++;
    p.tiles_m = (int)((p.M + BM - 1) / BM); p.tiles_n = (int)((D + BN - 1) / BN);
    p.kt_per_split = (int)(K / BK);
    return launch<EPI_PATCH_F32>((hipStream_t)stream, p, 1);
}
