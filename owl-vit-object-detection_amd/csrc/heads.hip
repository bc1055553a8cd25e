// Detection heads' tails.
//  * box_final : dense2 (D -> 4) + box bias + sigmoid + cxcywh -> xyxy
//                (HF5:998 dense2, HF5:1071-1104 bias, ref src/models.py:70-73).
//  * class_sims: e / (|e| + 1e-6) . (Q/|Q| + 1e-6)^T, MaxPool1d(3,3) over the query axis
//                (ref src/models.py:24-38; eps placement reproduced literally).  The [rows,Dt]x[Dt,32]
//                contraction runs on the exact-f32 matrix core (v_mfma_f32_32x32x2_f32) so the cosine
//                similarities carry f32 accuracy into the matcher / loss.
#include "common.h"

// ---- query bank normalisation: qhat[j,:] = Q[j,:]/|Q[j,:]| + 1e-6 ; rows >= nq zero ---------------
__global__ __launch_bounds__(64) void qhat_kernel(const float* __restrict__ q, float* qhat, float* qnorm, int nq, int Dt) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= nq) {
        for (int k = lane; k < Dt; k += 64) qhat[(int64_t)j * Dt + k] = 0.f;
        return;
    }
    float s = 0.f;
    for (int k = lane; k < Dt; k += 64) { const float v = q[(int64_t)j * Dt + k]; s += v * v; }
    const float n = sqrtf(wave_sum(s));
    if (lane == 0 && qnorm) qnorm[j] = n;
    for (int k = lane; k < Dt; k += 64) qhat[(int64_t)j * Dt + k] = q[(int64_t)j * Dt + k] / n + 1e-6f;
}

OWL_API int owl_query_normalize(void* stream, const float* queries, float* qhat32, float* qnorm, int64_t nq, int64_t Dt) {
    OWL_CHECK_ARG(queries && qhat32 && nq >= 1 && nq <= 32, "owl_query_normalize: need 1 <= queries <= 32 (got %lld)", (long long)nq);
    hipLaunchKernelGGL(qhat_kernel, dim3(32), dim3(64), 0, (hipStream_t)stream, queries, qhat32, qnorm, (int)nq, (int)Dt);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---- class sims ---------------------------------------------------------------------------------
// block = blockDim.x / 64 waves, each wave 32 rows.  qhat [32][Dt] f32 lives in LDS (row stride Dt+4 words).
// A lane's row is 2 KiB away from its neighbour's, so every load instruction touches 64 lines and the kernel lives on loads in flight:
// the e operand is fetched eight float4 (half of a lane's 64-float share of a 128-wide k chunk) AHEAD of the MFMAs that consume it,
// into two register blocks used alternately (round 6: 4 loads in flight per lane -> 8..16; arithmetic and its order unchanged, so the
// results are bit for bit those of the plain loop).  The host sizes the workgroup so that one workgroup per CU covers the rows.
__global__ __launch_bounds__(640) void class_sims_kernel(const float* __restrict__ e, const float* __restrict__ qhat,
                                                         float* sims, unsigned char* argmax, float* inv_norm,
                                                         int64_t rows, int Dt, int C) {
    extern __shared__ __attribute__((aligned(16))) float lq[];
    const int ldq = Dt + 4;
    const int NT = blockDim.x, nw = NT >> 6;
    float* lnorm = lq + 32 * ldq;   // [nw waves][32]
    for (int i = threadIdx.x; i < 32 * (Dt >> 2); i += NT) {
        const int j = i / (Dt >> 2), k4 = i - j * (Dt >> 2);
        *(float4*)(lq + j * ldq + k4 * 4) = ((const float4*)(qhat + (int64_t)j * Dt))[k4];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5;
    const int64_t r0 = ((int64_t)blockIdx.x * nw + w) * 32;
    int64_t row = r0 + (lane & 31);
    const bool valid = row < rows;
    if (!valid) row = rows - 1;
    const float* er = e + row * Dt;
    const float* qr = lq + (lane & 31) * ldq;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float ss = 0.f;
    // contraction split: lanes hi=0 take k in [c*128, c*128+64), hi=1 take [c*128+64, c*128+128); block = half of such a share (8 float4)
    const int nblk = 2 * ((Dt + 127) >> 7);
    auto kbase = [&](int blk, float& msk) {
        const int c0 = (blk >> 1) << 7;
        // no divergence around MFMAs: a half-chunk past Dt (Dt % 128 == 64) feeds zeros instead
        const bool act = (c0 + hi * 64) < Dt;
        msk = act ? 1.f : 0.f;
        return (act ? c0 + hi * 64 : 0) + (blk & 1) * 32;
    };
    auto fetch = [&](int blk, float4 (&buf)[8]) {
        float m;
        const float* p = er + kbase(blk, m);
#pragma unroll
        for (int s = 0; s < 8; s++) buf[s] = *(const float4*)(p + s * 4);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay AHEAD of the MFMAs of the other block (hipcc otherwise sinks them to their first use)
    };
    auto consume = [&](int blk, const float4 (&buf)[8]) {
        float msk;
        const float* q = qr + kbase(blk, msk);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            float4 a = buf[s];
            const float4 b = *(const float4*)(q + s * 4);
            a.x *= msk; a.y *= msk; a.z *= msk; a.w *= msk;
            {
                // separately rounded products and sums, in this order: the bits the kernel has always produced.  Left to -ffp-contract=fast hipcc fuses
                // these or not depending on the code AROUND them (the plain loop: not fused; this pipelined form: fused), and 1 / norm moves by an ulp.
#pragma clang fp contract(off)
                const float t = ((a.x * a.x + a.y * a.y) + a.z * a.z) + a.w * a.w;
                ss = ss + t;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    };
    float4 bufA[8], bufB[8];
    fetch(0, bufA);
    for (int blk = 0; blk < nblk; blk += 2) {       // nblk is even
        fetch(blk + 1, bufB);
        consume(blk, bufA);
        fetch(min(blk + 2, nblk - 1), bufA);        // unconditional (the last one re-reads a block and is dropped): a branch here makes hipcc wait for vmcnt(0)
        consume(blk + 1, bufB);
    }
    ss += __shfl_xor(ss, 32, 64);
    const float inv = 1.0f / (sqrtf(ss) + 1e-6f);
    if (hi == 0) {
        lnorm[w * 32 + lane] = inv;
        if (valid && inv_norm) inv_norm[row] = inv;
    }
    __syncthreads();
    // acc[r]: row i = (r&3) + 8*(r>>2) + 4*hi, query j = lane&31.  Max over query triples via shuffles.
    const int j = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v0 = acc[r] * lnorm[w * 32 + i];
        const float v1 = __shfl_down(v0, 1, 64), v2 = __shfl_down(v0, 2, 64);
        float best = v0; int arg = 0;
        if (v1 > best) { best = v1; arg = 1; }
        if (v2 > best) { best = v2; arg = 2; }
        const int64_t orow = r0 + i;
        if (j % 3 == 0 && j / 3 < C && orow < rows) {
            sims[orow * C + j / 3] = best;
            if (argmax) argmax[orow * C + j / 3] = (unsigned char)arg;
        }
    }
}

// waves per workgroup: one workgroup per CU when the rows allow it (B/16 batch 32: 2304 waves = 9 per CU exactly), four waves at small batch (batch 1: 18 workgroups; one-wave
// workgroups spend longer filling their 64 KiB query table than computing: 46.6 against 35.6 us from cold caches), at most 10 (two workgroups of 66 KiB LDS fit a CU)
static int class_sims_waves(int64_t rows) {
    const int64_t total = (rows + 31) / 32;
    const int64_t w = (total + 255) / 256;
    return (int)(w < 4 ? 4 : (w > 10 ? 10 : w));
}

OWL_API int owl_class_sims_fwd(void* stream, const float* e, const float* qhat32, float* sims, unsigned char* argmax,
                                  float* inv_norm, int64_t rows, int64_t Dt, int64_t C) {
    OWL_CHECK_ARG(e && qhat32 && sims, "owl_class_sims_fwd: null pointer");
    OWL_CHECK_ARG(rows >= 1, "owl_class_sims_fwd: rows >= 1 required");
    OWL_CHECK_ARG(Dt % 64 == 0 && 3 * C <= 32 && C >= 1, "owl_class_sims_fwd: Dt %% 64 == 0 and 3*C <= 32 required (Dt=%lld C=%lld)", (long long)Dt, (long long)C);
    const int nw = class_sims_waves(rows);
    const size_t shmem = (size_t)(32 * (Dt + 4) + 32 * nw) * sizeof(float);
    OWL_CHECK_ARG(shmem <= 160 * 1024, "owl_class_sims_fwd: Dt=%lld too large for the LDS-resident query table", (long long)Dt);
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, (void)hipFuncSetAttribute((const void*)class_sims_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t total = (rows + 31) / 32;
    hipLaunchKernelGGL(class_sims_kernel, dim3((unsigned)((total + nw - 1) / nw)), dim3(64 * nw), shmem, (hipStream_t)stream, e, qhat32, sims, argmax, inv_norm, rows, (int)Dt, (int)C);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---- box final ------------------------------------------------------------------------------------
// 4 dot products over D per row (bf16 activations, f32 weights), + bias + box_bias[p], sigmoid, cxcywh -> xyxy.  sig_out keeps sigma(.) for the
// backward.  Lane l owns elements 8 l .. 8 l + 7 (+ 512 for D > 512) of the row and adds its products in that order (fmaf chains); the four sums
// then meet across the wave pair by pair at distances 32, 16, ..., 1.
// A wave walks MANY rows (round 6; one wave per row before -- 54 -> 38 us, same bits): the lane's slice of the four weight rows lives in registers (it was re-read from
// the L1 for every row: 128 bytes per lane and row), the next two rows are requested while these two are worked on, and the four dot products meet in ONE
// butterfly (lanes trade halves of their four partial sums at distances 32 and 16, then four plain steps: 7 shuffles per row instead of 24) that ends with
// lanes 16 k .. 16 k + 15 holding output k -- each of those sums is formed pair by pair exactly as wave_sum forms it.
template <int NC>
__global__ __launch_bounds__(256) void box_final_rows_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, const float* __restrict__ box_bias,
                                                             float* boxes, float* sig_out, int64_t rows, int64_t P, int D, int rows_per_wave) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave;
    const int64_t row1 = min(rows, row0 + rows_per_wave);
    if (row0 >= row1) return;
    bool act[NC];
    float wr[NC][4][8];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int k = lane * 8 + 512 * c;
        act[c] = k < D;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float4 lo = make_float4(0, 0, 0, 0), hi4 = lo;
            if (act[c]) { lo = *(const float4*)(w2 + (int64_t)o * D + k); hi4 = *(const float4*)(w2 + (int64_t)o * D + k + 4); }
            wr[c][o][0] = lo.x; wr[c][o][1] = lo.y; wr[c][o][2] = lo.z; wr[c][o][3] = lo.w;
            wr[c][o][4] = hi4.x; wr[c][o][5] = hi4.y; wr[c][o][6] = hi4.z; wr[c][o][7] = hi4.w;
        }
    }
    const int grp = lane >> 4;                          // the output this lane ends up holding
    const float bias_k = b2[grp];
    struct RowIn { us8 hv[NC]; float bb; };
    auto fetch = [&](int64_t r, RowIn& x) {
        r = min(r, row1 - 1);                           // unconditional (the surplus fetches are dropped)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            x.hv[c] = us8{0, 0, 0, 0, 0, 0, 0, 0};
            if (act[c]) x.hv[c] = *(const us8*)(h + r * D + lane * 8 + 512 * c);
        }
        x.bb = box_bias[(r % P) * 4 + grp];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto process = [&](int64_t r, const RowIn& x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (act[c]) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float xv = bf2f(x.hv[c][e]);
                    a0 = fmaf(xv, wr[c][0][e], a0); a1 = fmaf(xv, wr[c][1][e], a1); a2 = fmaf(xv, wr[c][2][e], a2); a3 = fmaf(xv, wr[c][3][e], a3);
                }
            }
        }
        // distance 32: lanes < 32 go on with outputs (0, 1), the others with (2, 3); distance 16: bit 4 picks one of the pair
        const bool up = lane & 32;
        float v0 = up ? a2 : a0, v1 = up ? a3 : a1;
        v0 += __shfl_xor(up ? a0 : a2, 32, 64); v1 += __shfl_xor(up ? a1 : a3, 32, 64);
        const bool odd = lane & 16;
        float v = odd ? v1 : v0;
        v += __shfl_xor(odd ? v0 : v1, 16, 64);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const float sg = 1.f / (1.f + expf(-(v + bias_k + x.bb)));
        const float cx = __shfl(sg, 0, 64), cy = __shfl(sg, 16, 64), bw = __shfl(sg, 32, 64), bh = __shfl(sg, 48, 64);
        if (lane == 0) {
            if (sig_out) *(float4*)(sig_out + r * 4) = make_float4(cx, cy, bw, bh);
            *(float4*)(boxes + r * 4) = make_float4(cx - 0.5f * bw, cy - 0.5f * bh, cx + 0.5f * bw, cy + 0.5f * bh);
        }
    };
    RowIn x0, x1, x2;
    fetch(row0, x0); fetch(row0 + 1, x1);
    for (int64_t r = row0; r < row1; r += 3) {
        fetch(r + 2, x2); process(r, x0);
        if (r + 1 >= row1) break;
        fetch(r + 3, x0); process(r + 1, x1);
        if (r + 2 >= row1) break;
        fetch(r + 4, x1); process(r + 2, x2);
    }
}

OWL_API int owl_box_final_fwd(void* stream, const void* h_bf16, const float* w2, const float* b2, const float* box_bias,
                                 float* boxes, float* sig_out, int64_t rows, int64_t P, int64_t D) {
    OWL_CHECK_ARG(h_bf16 && w2 && b2 && box_bias && boxes, "owl_box_final_fwd: null pointer");
    OWL_CHECK_ARG(D % 8 == 0 && D >= 8 && D <= 1024, "owl_box_final_fwd: D %% 8 == 0 and D <= 1024 required (the LayerNorm kernels' own limit)");
    // rows per wave: 16 at large batch (the weight slice is loaded once per wave), fewer when that would leave CUs idle (batch 1: 2304 rows -> 1)
    int64_t rpw = (rows + 4095) / 4096;
    rpw = rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
    const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
    if (D <= 512)
        hipLaunchKernelGGL(box_final_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h_bf16, w2, b2, box_bias, boxes, sig_out, rows, P, (int)D, (int)rpw);
    else
        hipLaunchKernelGGL(box_final_rows_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h_bf16, w2, b2, box_bias, boxes, sig_out, rows, P, (int)D, (int)rpw);
    OWL_LAUNCH_CHECK();
    return 0;
}
