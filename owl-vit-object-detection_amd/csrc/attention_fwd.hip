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

struct AttnFwdP {
    const bf16_t* q; const bf16_t* k; int64_t ld_qk;   // row-major [B*Tp, ld]; head h at column h*64
    const bf16_t* vt; int64_t vt_img_stride;            // V^T [B][heads..][64][Tp]; element stride per image
    bf16_t* out; int64_t ld_out;                        // [B*Tp, ld_out], head h at column h*64
    float* lse;                                         // optional [B][H][Tp], log2 domain
    int T, Tp, H;
    float scale_log2e;
};

__device__ __forceinline__ int swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnFwdP p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 16384];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + w * 32;
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
    const bf16_t* vbase = p.vt + (int64_t)b * p.vt_img_stride + (int64_t)h * 64 * p.Tp;
    auto stage = [&](int buf, int kv) {
        unsigned char* base = lds + buf * 16384;
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
            const int r0 = (w * 2 + qd) * 8;
            const int r = r0 + (lane >> 3);
            const int ch = (lane & 7) ^ ((r >> 1) & 7);
            int key = kv * 64 + r;
            if (key >= p.T) key = p.T - 1;
            __builtin_amdgcn_global_load_lds(GPTR(kbase + (int64_t)key * p.ld_qk + ch * 8), LPTR(base + r0 * 128), 16, 0, 0);
            // V^T row r = d; 8 keys per chunk (reads past T are finite junk, masked by P = 0)
            __builtin_amdgcn_global_load_lds(GPTR(vbase + (int64_t)r * p.Tp + kv * 64 + ch * 8), LPTR(base + 8192 + r0 * 128), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    // fragment addressing
    int k_off[2], k_sw[2], v_off[2], v_sw[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int rk = t * 32 + swap23(lane & 31);
        k_off[t] = rk * 128; k_sw[t] = (rk >> 1) & 7;
        const int rv = t * 32 + (lane & 31);
        v_off[t] = 8192 + rv * 128; v_sw[t] = (rv >> 1) & 7;
    }

    const int nkv = (p.T + 63) / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;
    for (int kv = 0; kv < nkv; kv++) {
        if (kv + 1 < nkv) stage(cur ^ 1, kv + 1);
        const unsigned char* tb = lds + cur * 16384;

        // ---- S^T = K Q^T : two 32-key tiles ----------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) s[t][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const bf16x8 kf = *(const bf16x8*)(tb + k_off[t] + (((kc * 2 + hi) ^ k_sw[t]) << 4));
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kc], s[t], 0, 0, 0);
            }
        }
        // lane's key for register r of tile t:  kv*64 + t*32 + 16*(r>>3) + 8*hi + (r&7)
        if (kv * 64 + 64 > p.T) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = kv * 64 + t * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= p.T) s[t][r] = -INFINITY;
                }
        }
        // ---- online softmax (per query column = per lane pair {l, l^32}) -------------------------
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c, -mc));
                s[t][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[d][r] *= alpha;

        // ---- O^T += V^T P^T : 4 chunks of 16 keys, 2 d-blocks ------------------------------------
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; j++) pf[j] = (short)f2bf(s[t][cc * 8 + j]);
                const int ch = (t * 2 + cc) * 2 + hi;
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const bf16x8 vf = *(const bf16x8*)(tb + v_off[d] + ((ch ^ v_sw[d]) << 4));
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
    }

    // ---- epilogue: normalise, write O (row-major, head h) and LSE ---------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + (lane & 31);
    if (qr < p.T) {
        bf16_t* op = p.out + ((int64_t)b * p.Tp + qr) * p.ld_out + h * 64;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                uint2 v;
                v.x = pack_bf2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
                v.y = pack_bf2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
                *(uint2*)(op + d * 32 + 8 * qd + 4 * hi) = v;
            }
        if (p.lse && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Tp + qr] = m_run * c + __builtin_amdgcn_logf(l_tot);
    }
}

extern "C" int owl_attention_fwd_bf16(void* stream, const void* q, const void* k, int64_t ld_qk, const void* vt,
                                      int64_t vt_img_stride, void* out, int64_t ld_out, float* lse, int64_t B,
                                      int64_t H, int64_t T, int64_t Tp, float scale) {
    OWL_CHECK_ARG(q && k && vt && out, "owl_attention_fwd_bf16: null pointer");
    OWL_CHECK_ARG(ld_qk % 8 == 0 && ld_out % 4 == 0 && Tp % 8 == 0 && T > 0 && T <= Tp, "owl_attention_fwd_bf16: bad strides (ld_qk %% 8, ld_out %% 4, Tp %% 8)");
    AttnFwdP p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.ld_qk = ld_qk;
    p.vt = (const bf16_t*)vt; p.vt_img_stride = vt_img_stride;
    p.out = (bf16_t*)out; p.ld_out = ld_out; p.lse = lse;
    p.T = (int)T; p.Tp = (int)Tp; p.H = (int)H;
    p.scale_log2e = scale * 1.4426950408889634f;
    dim3 grid((unsigned)((T + 127) / 128), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    OWL_LAUNCH_CHECK();
    return 0;
}
