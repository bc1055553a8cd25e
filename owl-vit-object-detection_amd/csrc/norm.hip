// LayerNorm family (HBM-bound; one wave per row, float4 loads, row kept in registers).
// Reference: nn.LayerNorm eps 1e-5 at HF5:484-486 (layer_norm1/2), HF5:721-723 (pre/post),
// HF5:1060 (`layer_norm` = ref post_post_layernorm); class-token merge ref src/models.py:80-86.
#include "common.h"

static constexpr int LN_MAXV = 4;  // float4 per lane -> D <= 1024

template <int NV>
__device__ __forceinline__ void row_stats(const float4 (&v)[NV], int nvec, int lane, int D, float& mean, float& rstd, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
        if (lane + i * 64 < nvec) s += v[i].x + v[i].y + v[i].z + v[i].w;
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
        if (lane + i * 64 < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

// y = LN(x [+ delta]) * gamma + beta ; out bf16 (OUT_BF16) or f32 (may alias x); stats[row] = {mean, rstd}.
// With `delta` (bf16 GEMM output of the previous residual branch) the residual add is fused here:
// x_new = x + delta is written to x_out (f32, may alias x) -- the GEMM epilogue then stores 2 bytes per
// element instead of reading and writing 4 (the f32 residual epilogue was HBM/issue-bound).
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, const bf16_t* __restrict__ delta, float* x_out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, void* out,
                                                     float2* stats, int64_t rows, int D, float eps, const bf16_t* __restrict__ delta2) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = D >> 2;
    const float4* xr = (const float4*)(x + row * D);
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (lane + i * 64 < nvec) {
            // x and the branch outputs are dead once this kernel has read them: streaming loads (common.h)
            v[i] = ld_stream_f4((const float*)(xr + lane + i * 64));
            if (delta) {
                const uint2 u = ld_stream_u2((const uint2*)(delta + row * D) + lane + i * 64);
                v[i].x += bf2f(u.x & 0xffff); v[i].y += bf2f(u.x >> 16); v[i].z += bf2f(u.y & 0xffff); v[i].w += bf2f(u.y >> 16);
                if (delta2) {                       // (x + delta) + delta2: the two branch outputs of a layer whose first add was not stored
                    const uint2 u2 = ld_stream_u2((const uint2*)(delta2 + row * D) + lane + i * 64);
                    v[i].x += bf2f(u2.x & 0xffff); v[i].y += bf2f(u2.x >> 16); v[i].z += bf2f(u2.y & 0xffff); v[i].w += bf2f(u2.y >> 16);
                }
                // The sum is not read again before the next LayerNorm (a QKV GEMM, an attention and an out-proj launch later), while h -- written
                // by this same kernel -- is the next GEMM's A operand: streaming store.
                if (x_out) st_stream_f4(x_out + row * D + 4 * (lane + i * 64), v[i]);
            }
        }
    float mean, rstd;
    row_stats<LN_MAXV>(v, nvec, lane, D, mean, rstd, eps);
    if (stats && lane == 0) stats[row] = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 g = ((const float4*)gamma)[idx], bb = ((const float4*)beta)[idx];
            const float y0 = (v[i].x - mean) * rstd * g.x + bb.x, y1 = (v[i].y - mean) * rstd * g.y + bb.y;
            const float y2 = (v[i].z - mean) * rstd * g.z + bb.z, y3 = (v[i].w - mean) * rstd * g.w + bb.w;
            if constexpr (OUT_BF16) {
                uint2 o; o.x = pack_bf2(y0, y1); o.y = pack_bf2(y2, y3);
                ((uint2*)((bf16_t*)out + row * D))[idx] = o;
            } else {
                ((float4*)((float*)out + row * D))[idx] = make_float4(y0, y1, y2, y3);
            }
        }
    }
}

static int ln_launch(void* stream, const float* x, const void* delta, float* x_out, const float* gamma, const float* beta, void* out,
                     int out_bf16, float* stats, int64_t rows, int64_t D, float eps, const void* delta2 = nullptr) {
    OWL_CHECK_ARG(x && gamma && beta && out, "owl_layernorm_fwd: null pointer");
    OWL_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV, "owl_layernorm_fwd: D=%lld must be a multiple of 4 and <= 1024", (long long)D);
    OWL_CHECK_ARG(delta || !delta2, "owl_add_layernorm_fwd: delta2 without delta");
    dim3 grid((unsigned)((rows + 3) / 4));
    if (out_bf16)
        hipLaunchKernelGGL(ln_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)delta, x_out, gamma, beta, out, (float2*)stats, rows, (int)D, eps, (const bf16_t*)delta2);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)delta, x_out, gamma, beta, out, (float2*)stats, rows, (int)D, eps, (const bf16_t*)delta2);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_layernorm_fwd(void* stream, const float* x, const float* gamma, const float* beta, void* out,
                                 int out_bf16, float* stats, int64_t rows, int64_t D, float eps) {
    return ln_launch(stream, x, nullptr, nullptr, gamma, beta, out, out_bf16, stats, rows, D, eps);
}

// s = x + delta (+ delta2), bf16 branch outputs;  out = LN(s);  x_out = s unless x_out is null (the sum is then re-formed, from the
// same operands in the same order, by the next call -- saves writing 4 bytes per element where nobody else reads the sum)
OWL_API int owl_add_layernorm_fwd(void* stream, const float* x, const void* delta_bf16, float* x_out, const float* gamma,
                                     const float* beta, void* out, int out_bf16, float* stats, int64_t rows, int64_t D, float eps,
                                     const void* delta2_bf16) {
    OWL_CHECK_ARG(delta_bf16, "owl_add_layernorm_fwd: null delta");
    return ln_launch(stream, x, delta_bf16, x_out, gamma, beta, out, out_bf16, stats, rows, D, eps, delta2_bf16);
}

// Class-token rows: X[b*Tp + 0, :] = class_embedding + pos[0]   (HF5:338-343)
__global__ void cls_rows_kernel(float* x, const float* cls, const float* pos, int64_t B, int64_t Tp, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int64_t b = i / D;
    const int d = (int)(i - b * D);
    x[b * Tp * D + d] = cls[d] + pos[d];
}

OWL_API int owl_cls_rows(void* stream, float* x, const float* cls, const float* pos, int64_t B, int64_t Tp, int64_t D) {
    OWL_CHECK_ARG(x && cls && pos, "owl_cls_rows: null pointer");
    hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)((B * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, cls, pos, B, Tp, (int)D);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---- class-token merge + second LN (ref src/models.py:80-86) -------------------------------------
// cls_ln[b,:] = LN1(X[b,0,:]) ;  feats[b*P + p, :] = LN2( LN1(X[b,1+p,:]) * cls_ln[b,:] )
// stats1[b*Tp + t] = (mean, rstd) of LN1 on token t (t = 0 written by the cls pass), stats2[b*P+p] of LN2.
__global__ __launch_bounds__(256) void merge_ln_kernel(const float* x, const bf16_t* __restrict__ delta, float* x_out, const float* __restrict__ cls_ln,
                                                       const float* __restrict__ g1, const float* __restrict__ b1,
                                                       const float* __restrict__ g2, const float* __restrict__ b2,
                                                       bf16_t* feats, float2* stats1, float2* stats2, int64_t B,
                                                       int64_t P, int64_t Tp, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // b*P + p
    if (row >= B * P) return;
    const int64_t b = row / P, pp = row - b * P;
    const int nvec = D >> 2;
    const int64_t xrow = b * Tp + 1 + pp;
    const float4* xr = (const float4*)(x + xrow * D);
    const float4* cr = (const float4*)(cls_ln + b * D);
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (lane + i * 64 < nvec) {
            v[i] = ld_stream_f4((const float*)(xr + lane + i * 64));
            if (delta) {
                const uint2 u = ld_stream_u2((const uint2*)(delta + xrow * D) + lane + i * 64);
                v[i].x += bf2f(u.x & 0xffff); v[i].y += bf2f(u.x >> 16); v[i].z += bf2f(u.y & 0xffff); v[i].w += bf2f(u.y >> 16);
                st_stream_f4(x_out + xrow * D + 4 * (lane + i * 64), v[i]);          // (read again by the backward only)
            }
        }
    float mean, rstd;
    row_stats<LN_MAXV>(v, nvec, lane, D, mean, rstd, eps);
    if (stats1 && lane == 0) stats1[b * Tp + 1 + pp] = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 g = ((const float4*)g1)[idx], bb = ((const float4*)b1)[idx], c = cr[idx];
            v[i].x = ((v[i].x - mean) * rstd * g.x + bb.x) * c.x;
            v[i].y = ((v[i].y - mean) * rstd * g.y + bb.y) * c.y;
            v[i].z = ((v[i].z - mean) * rstd * g.z + bb.z) * c.z;
            v[i].w = ((v[i].w - mean) * rstd * g.w + bb.w) * c.w;
        }
    }
    float mean2, rstd2;
    row_stats<LN_MAXV>(v, nvec, lane, D, mean2, rstd2, eps);
    if (stats2 && lane == 0) stats2[row] = make_float2(mean2, rstd2);
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 g = ((const float4*)g2)[idx], bb = ((const float4*)b2)[idx];
            uint2 o;
            o.x = pack_bf2((v[i].x - mean2) * rstd2 * g.x + bb.x, (v[i].y - mean2) * rstd2 * g.y + bb.y);
            o.y = pack_bf2((v[i].z - mean2) * rstd2 * g.z + bb.z, (v[i].w - mean2) * rstd2 * g.w + bb.w);
            ((uint2*)(feats + row * D))[idx] = o;
        }
    }
}

// cls pass: one wave per image
__global__ __launch_bounds__(64) void cls_ln_kernel(const float* x, const bf16_t* __restrict__ delta, float* x_out, const float* __restrict__ g1,
                                                    const float* __restrict__ b1, float* cls_ln, float2* stats1,
                                                    int64_t Tp, int D, float eps) {
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int nvec = D >> 2;
    const float4* xr = (const float4*)(x + b * Tp * D);
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (lane + i * 64 < nvec) {
            v[i] = xr[lane + i * 64];
            if (delta) {
                const uint2 u = ((const uint2*)(delta + b * Tp * D))[lane + i * 64];
                v[i].x += bf2f(u.x & 0xffff); v[i].y += bf2f(u.x >> 16); v[i].z += bf2f(u.y & 0xffff); v[i].w += bf2f(u.y >> 16);
                ((float4*)(x_out + b * Tp * D))[lane + i * 64] = v[i];
            }
        }
    float mean, rstd;
    row_stats<LN_MAXV>(v, nvec, lane, D, mean, rstd, eps);
    if (stats1 && lane == 0) stats1[b * Tp] = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 g = ((const float4*)g1)[idx], bb = ((const float4*)b1)[idx];
            ((float4*)(cls_ln + b * D))[idx] = make_float4((v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y,
                                                           (v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w);
        }
    }
}

OWL_API int owl_merge_ln_fwd(void* stream, const float* x, const void* delta_bf16, float* x_out, const float* g1, const float* b1,
                                const float* g2, const float* b2, float* cls_ln, void* feats_bf16, float* stats1, float* stats2,
                                int64_t B, int64_t P, int64_t Tp, int64_t D, float eps) {
    OWL_CHECK_ARG(x && g1 && b1 && g2 && b2 && cls_ln && feats_bf16, "owl_merge_ln_fwd: null pointer");
    OWL_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV, "owl_merge_ln_fwd: D must be a multiple of 4 and <= 1024");
    OWL_CHECK_ARG((delta_bf16 == nullptr) || (x_out != nullptr), "owl_merge_ln_fwd: x_out required with delta");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cls_ln_kernel, dim3((unsigned)B), dim3(64), 0, s, x, (const bf16_t*)delta_bf16, x_out, g1, b1, cls_ln, (float2*)stats1, Tp, (int)D, eps);
    OWL_LAUNCH_CHECK();
    hipLaunchKernelGGL(merge_ln_kernel, dim3((unsigned)((B * P + 3) / 4)), dim3(256), 0, s, x, (const bf16_t*)delta_bf16, x_out, cls_ln, g1, b1, g2, b2,
                       (bf16_t*)feats_bf16, (float2*)stats1, (float2*)stats2, B, P, Tp, (int)D, eps);
    OWL_LAUNCH_CHECK();
    return 0;
}
