// Backward kernels for the row-wise (HBM-bound) pieces of the train path: LayerNorm, class-token merge,
// class head, box head tail, bias gradients.  The GEMM-shaped backward work (dX, dW) reuses gemm.hip.
// Reference: these are the autograd forms of ref src/models.py:24-38, 65-73, 80-86 and HF5:484-509.
// Parameter-gradient reductions over rows are DETERMINISTIC: every workgroup writes its partial sums to a caller-provided
// f32 workspace and a second kernel adds them in a fixed order into the flat gradient bucket (no f32 atomics anywhere:
// the bucket is bitwise reproducible run to run, tests/test_determinism_gpu.py).
#include "common.h"

static constexpr int LN_MAXV = 4;

// out_k[c] (+)= sum_{s < nblk} part[g*group_stride + s*stride + k*seg + c]   for k < nseg, c < seg, every group g = blockIdx.y
// (out_k advanced by g*out_group_stride).  Fixed summation order: a thread adds slabs s = j, j+16, ... in order, then the 16
// partial sums of a column are added in order 0..15.  seg % 4 == 0.
struct ReduceOuts { float* o[5]; };
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, ReduceOuts outs, int nseg, int seg, int64_t stride,
                                                              int nblk, int64_t group_stride, int64_t out_group_stride, int accumulate) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, j = threadIdx.x >> 4;          // 16 column quads x 16 slab lanes
    const int64_t n = (int64_t)nseg * seg;
    const int64_t c = ((int64_t)blockIdx.x * 16 + cq) * 4;
    const float* base = part + (int64_t)blockIdx.y * group_stride + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n) {
#pragma unroll 4
        for (int s = j; s < nblk; s += 16) {
            const float4 v = *(const float4*)(base + (int64_t)s * stride);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[j][cq] = a;
    __syncthreads();
    if (j == 0 && c < n) {
        float4 t = red[0][cq];
#pragma unroll
        for (int k = 1; k < 16; k++) { const float4 v = red[k][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const int k = (int)(c / seg);
        float* o = outs.o[k] + (int64_t)blockIdx.y * out_group_stride + (c - (int64_t)k * seg);
        if (accumulate) { const float4 r = *(const float4*)o; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
        *(float4*)o = t;
    }
}

static int partials_reduce(hipStream_t s, const float* part, ReduceOuts outs, int nseg, int seg, int64_t stride, int nblk, int groups,
                           int64_t group_stride, int64_t out_group_stride, int accumulate) {
    const int64_t n = (int64_t)nseg * seg;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((unsigned)((n / 4 + 15) / 16), (unsigned)groups), dim3(256), 0, s, part, outs, nseg, seg, stride,
                       nblk, group_stride, out_group_stride, accumulate);
    OWL_LAUNCH_CHECK();
    return 0;
}

// f32 elements of partial-sum scratch that serve every row-reduction entry below for activations of `groups` images x
// `rows_per_group` rows x up to C columns (C = the widest reduced matrix, e.g. the MLP width for the fc1 bias gradient)
OWL_API int owl_rowreduce_workspace_bytes(int64_t groups, int64_t rows_per_group, int64_t C, int64_t* bytes) {
    OWL_CHECK_ARG(bytes && groups >= 1 && rows_per_group >= 1 && C >= 4, "owl_rowreduce_workspace_bytes: bad arguments");
    *bytes = groups * ((rows_per_group + 63) / 64) * 5 * C * (int64_t)sizeof(float);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: dx = (dres) + rstd * (g*dy - mean(g*dy) - xhat*mean(g*dy*xhat));  dgamma += dy*xhat,
// dbeta += dy.  dy is bf16 (GEMM output) or f32; one wave per row, RPB rows per workgroup.
// DXSUM: also the column sums of dx (= the bias gradient of the linear layer whose output this residual position is: saves that
// layer's separate column-sum pass over the f32 dx).
// ---------------------------------------------------------------------------------------------------
// NV: D = 256 NV exactly (the model widths: 3 = 768, 4 = 1024; 0 = any D <= 1024 with per-lane bounds).  Round 6: at D = 768 the accumulators and the
// LDS reduction buffers are sized for 3 vectors per lane, not 4 (116 -> 121 registers with the hoisted row below, 33 -> 25 KiB: four workgroups per CU under
// ln_bwd_kernel_768's bound; the column-sum form needs 133 and runs at three waves per SIMD without it), the residual gradient's row is requested with
// the row's other operands instead of behind the two wave reductions (one load phase per row, not two), and HAS_DX = false (parameter gradients only:
// the trainable layer's LN1) drops everything the dx path keeps alive (66 registers).
template <bool DY_BF16, bool DXSUM, int NV, bool HAS_DX>
__device__ __forceinline__ void ln_bwd_body(const void* __restrict__ dy_, const float* __restrict__ x,
                                            const float2* __restrict__ stats, const float* __restrict__ gamma,
                                            const float* dres, float* dx, float* part, int64_t rows,
                                            int D, int rows_per_block, bf16_t* dx_bf16, float (&red)[DXSUM ? 3 : 2][4][(NV ? NV : LN_MAXV) * 256 + 4]) {
    // Every product-sum below is written out (fmaf where one rounding is meant, separate operations elsewhere) with contraction off: the three
    // instantiations per form (any D / 768 / 1024) and any later compiler then produce the same bits -- left to -ffp-contract=fast hipcc picks a
    // different fusion per instantiation (which product of a sum it folds depends on the basic-block structure around it).
#pragma clang fp contract(off)
    constexpr int V = NV ? NV : LN_MAXV;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nvec = D >> 2;
    float4 ag[V], ab[V], ad[DXSUM ? V : 1];
#pragma unroll
    for (int i = 0; i < V; i++) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); if (DXSUM) ad[DXSUM ? i : 0] = make_float4(0, 0, 0, 0); }
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = min(rows, r_begin + rows_per_block);
    const bool with_res = HAS_DX && dres;                      // (wave-uniform)
    for (int64_t row = r_begin + w; row < r_end; row += 4) {
        const float2 st = stats[row];
        float4 xh[V], gd[V], rs[HAS_DX ? V : 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; i++) {
            const int idx = lane + i * 64;
            if constexpr (HAS_DX) rs[i] = make_float4(0, 0, 0, 0);
            if (NV || idx < nvec) {
                const float4 xv = ld_stream_f4(x + row * D + 4 * idx);          // saved activation, upstream gradient, residual gradient: all dead after this
                float4 dyv;
                if constexpr (DY_BF16) {
                    const uint2 u = ld_stream_u2((const uint2*)((const bf16_t*)dy_ + row * D) + idx);
                    dyv = make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
                } else {
                    dyv = ld_stream_f4((const float*)dy_ + row * D + 4 * idx);
                }
                if constexpr (HAS_DX) { if (with_res) rs[i] = ld_stream_f4(dres + row * D + 4 * idx); }
                const float4 g = ((const float4*)gamma)[idx];
                xh[i] = make_float4((xv.x - st.x) * st.y, (xv.y - st.x) * st.y, (xv.z - st.x) * st.y, (xv.w - st.x) * st.y);
                gd[i] = make_float4(dyv.x * g.x, dyv.y * g.y, dyv.z * g.z, dyv.w * g.w);
                s1 += ((gd[i].x + gd[i].y) + gd[i].z) + gd[i].w;
                s2 += fmaf(gd[i].w, xh[i].w, fmaf(gd[i].z, xh[i].z, fmaf(gd[i].y, xh[i].y, gd[i].x * xh[i].x)));
                ag[i].x = fmaf(dyv.x, xh[i].x, ag[i].x); ag[i].y = fmaf(dyv.y, xh[i].y, ag[i].y);
                ag[i].z = fmaf(dyv.z, xh[i].z, ag[i].z); ag[i].w = fmaf(dyv.w, xh[i].w, ag[i].w);
                ab[i].x += dyv.x; ab[i].y += dyv.y; ab[i].z += dyv.z; ab[i].w += dyv.w;
            }
        }
        s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
        if constexpr (HAS_DX) {
#pragma unroll
            for (int i = 0; i < V; i++) {
                const int idx = lane + i * 64;
                if (NV || idx < nvec) {
                    float4 o = make_float4(st.y * fmaf(-xh[i].x, s2, gd[i].x - s1), st.y * fmaf(-xh[i].y, s2, gd[i].y - s1),
                                           st.y * fmaf(-xh[i].z, s2, gd[i].z - s1), st.y * fmaf(-xh[i].w, s2, gd[i].w - s1));
                    if (dres) {
                        const float4 r = rs[HAS_DX ? i : 0];
                        o.x = o.x + r.x; o.y = o.y + r.y; o.z = o.z + r.z; o.w = o.w + r.w;
                    }
                    st_stream_f4(dx + row * D + 4 * idx, o);          // f32 gradient stream: next read by the LayerNorm backward two GEMMs later
                    if constexpr (DXSUM) { float4& a = ad[DXSUM ? i : 0]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
                    if (dx_bf16) {                          // the bf16 copy the next dX GEMM reads (saves a separate cast pass)
                        uint2 ob; ob.x = pack_bf2(o.x, o.y); ob.y = pack_bf2(o.z, o.w);
                        ((uint2*)(dx_bf16 + row * D))[idx] = ob;
                    }
                }
            }
        }
    }
    if (!part) return;
    // reduce the 4 waves' partials through LDS; this workgroup's sums go to part[blockIdx.x][{dgamma, dbeta}][D]
#pragma unroll
    for (int i = 0; i < V; i++) {
        const int idx = lane + i * 64;
        if (NV || idx < nvec) {
            *(float4*)&red[0][w][idx * 4] = ag[i]; *(float4*)&red[1][w][idx * 4] = ab[i];
            if constexpr (DXSUM) *(float4*)&red[2][w][idx * 4] = ad[DXSUM ? i : 0];
        }
    }
    __syncthreads();
    constexpr int NS = DXSUM ? 3 : 2;
    float* mine = part + (int64_t)blockIdx.x * NS * D;
    for (int c = threadIdx.x; c < D; c += 256) {
#pragma unroll
        for (int k = 0; k < NS; k++) mine[k * D + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
    }
}

template <bool DY_BF16, bool DXSUM, int NV, bool HAS_DX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x, const float2* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* dres, float* dx, float* part, int64_t rows,
                                                     int D, int rows_per_block, bf16_t* dx_bf16) {
    __shared__ float red[DXSUM ? 3 : 2][4][(NV ? NV : LN_MAXV) * 256 + 4];
    ln_bwd_body<DY_BF16, DXSUM, NV, HAS_DX>(dy_, x, stats, gamma, dres, dx, part, rows, D, rows_per_block, dx_bf16, red);
}
// D = 768: four waves per SIMD
template <bool DY_BF16, bool DXSUM, bool HAS_DX>
__global__ __launch_bounds__(256, 4) void ln_bwd_kernel_768(const void* __restrict__ dy_, const float* __restrict__ x, const float2* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* dres, float* dx, float* part, int64_t rows,
                                                            int D, int rows_per_block, bf16_t* dx_bf16) {
    __shared__ float red[DXSUM ? 3 : 2][4][3 * 256 + 4];
    ln_bwd_body<DY_BF16, DXSUM, 3, HAS_DX>(dy_, x, stats, gamma, dres, dx, part, rows, D, rows_per_block, dx_bf16, red);
}

template <bool DY_BF16, bool DXSUM, bool HAS_DX>
static void ln_bwd_launch2(dim3 grid, hipStream_t s, const void* dy, const float* x, const float2* stats, const float* gamma, const float* dres, float* dx,
                           float* part, int64_t rows, int D, int rpb, bf16_t* dx_bf16) {
    if (D == 768) {
        // (the column-sum form needs 134 registers: bounded to four waves per SIMD it spills 5 -- it runs at three, without the bound)
        if constexpr (DXSUM) hipLaunchKernelGGL((ln_bwd_kernel<DY_BF16, true, 3, HAS_DX>), grid, dim3(256), 0, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);
        else hipLaunchKernelGGL((ln_bwd_kernel_768<DY_BF16, false, HAS_DX>), grid, dim3(256), 0, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);
    }
    else if (D == 1024) hipLaunchKernelGGL((ln_bwd_kernel<DY_BF16, DXSUM, 4, HAS_DX>), grid, dim3(256), 0, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);
    else hipLaunchKernelGGL((ln_bwd_kernel<DY_BF16, DXSUM, 0, HAS_DX>), grid, dim3(256), 0, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);
}
template <bool DY_BF16, bool DXSUM>
static void ln_bwd_launch(dim3 grid, hipStream_t s, const void* dy, const float* x, const float2* stats, const float* gamma, const float* dres, float* dx,
                          float* part, int64_t rows, int D, int rpb, bf16_t* dx_bf16) {
    if (dx) ln_bwd_launch2<DY_BF16, DXSUM, true>(grid, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);
    else if constexpr (!DXSUM) ln_bwd_launch2<DY_BF16, false, false>(grid, s, dy, x, stats, gamma, dres, dx, part, rows, D, rpb, dx_bf16);     // (column sums of dx come with dx)
}

OWL_API int owl_layernorm_bwd(void* stream, const void* dy, int dy_bf16, const float* x, const float* stats, const float* gamma,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D, void* dx_bf16,
                                 float* partials, int64_t partials_floats, float* dx_colsum) {
    OWL_CHECK_ARG(dy && x && stats && gamma, "owl_layernorm_bwd: null pointer");
    OWL_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV, "owl_layernorm_bwd: D must be a multiple of 4 and <= 1024");
    OWL_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "owl_layernorm_bwd: dgamma/dbeta both or neither");
    OWL_CHECK_ARG(!dx_bf16 || dx, "owl_layernorm_bwd: dx_bf16 needs dx");
    const int rpb = 64;
    const int nblk = (int)((rows + rpb - 1) / rpb);
    dim3 grid((unsigned)nblk);
    float* part = nullptr;
    OWL_CHECK_ARG(!dx_colsum || (dgamma && dx && dy_bf16), "owl_layernorm_bwd: dx_colsum comes with dgamma / dbeta, dx and a bf16 dy");
    const int ns = dx_colsum ? 3 : 2;
    if (dgamma) {
        OWL_CHECK_ARG(partials && partials_floats >= (int64_t)nblk * ns * D, "owl_layernorm_bwd: parameter gradients need %lld floats of partial-sum scratch (owl_rowreduce_workspace_bytes)", (long long)nblk * ns * D);
        part = partials;
    }
    if (dx_colsum) ln_bwd_launch<true, true>(grid, (hipStream_t)stream, dy, x, (const float2*)stats, gamma, dres, dx, part, rows, (int)D, rpb, (bf16_t*)dx_bf16);
    else if (dy_bf16) ln_bwd_launch<true, false>(grid, (hipStream_t)stream, dy, x, (const float2*)stats, gamma, dres, dx, part, rows, (int)D, rpb, (bf16_t*)dx_bf16);
    else ln_bwd_launch<false, false>(grid, (hipStream_t)stream, dy, x, (const float2*)stats, gamma, dres, dx, part, rows, (int)D, rpb, (bf16_t*)dx_bf16);
    OWL_LAUNCH_CHECK();
    if (!part) return 0;
    ReduceOuts outs{}; outs.o[0] = dgamma; outs.o[1] = dbeta; outs.o[2] = dx_colsum;
    return partials_reduce((hipStream_t)stream, part, outs, ns, (int)D, (int64_t)ns * D, nblk, 1, 0, 0, 1);
}

// ---------------------------------------------------------------------------------------------------
// merge + LN2 backward (ref src/models.py:80-86).  Patch rows: from d_feats (f32) back to the residual
// stream rows 1..P of each image, accumulating d(cls_ln)[b,:] and the four LN parameter gradients.
// grid = (ceil(P / RPB), B).
// ---------------------------------------------------------------------------------------------------
// A THREAD owns four columns for the whole workgroup (D <= 1024): its slices of the four parameter vectors and of cls_ln are loaded once, its
// five column sums need no cross-wave reduction, and the only traffic per row is the row itself (x, dfeats in; dx f32 + bf16 out).  The
// workgroup walks its rows MLR at a time: the two pairs of row moments go through LDS (wave sums -> [wave][row] -> every thread adds the four
// waves' parts in wave order), two barriers per MLR rows.  (The wave-per-row form this replaces re-read seven parameter vectors per row --
// 78 % of its load instructions -- and needed 244 registers for its 5 x 16 accumulators: 238 us at the headline size, this one 190.)
static constexpr int MLR = 2;
__global__ __launch_bounds__(256) void merge_ln_bwd_kernel(const float* __restrict__ dfeats, const float* __restrict__ x,
                                                           const float* __restrict__ cls_ln, const float2* __restrict__ stats1,
                                                           const float2* __restrict__ stats2, const float* __restrict__ g1,
                                                           const float* __restrict__ b1, const float* __restrict__ g2, float* __restrict__ dx,
                                                           float* __restrict__ part, int64_t P, int64_t Tp, int D, int rows_per_block,
                                                           bf16_t* __restrict__ dx_bf16) {
    __shared__ float2 redm[2][4][MLR];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t b = blockIdx.y;
    const bool ok = 4 * t < D;
    const float4 z4 = make_float4(0, 0, 0, 0);
    float4 ga = z4, be = z4, gb = z4, cv = z4;
    if (ok) { ga = ((const float4*)g1)[t]; be = ((const float4*)b1)[t]; gb = ((const float4*)g2)[t]; cv = ((const float4*)(cls_ln + b * D))[t]; }
    float4 a_c = z4, a_g1 = z4, a_b1 = z4, a_g2 = z4, a_b2 = z4, a_dx = z4;      // (a_dx: column sums of dx = the bias gradient of the last fc2)
    const float invD = 1.0f / (float)D;
    const int64_t p_begin = (int64_t)blockIdx.x * rows_per_block, p_end = min(P, p_begin + rows_per_block);
    // the NEXT MLR rows are requested before these are worked on: a workgroup's loads stay in flight through its compute and barrier phases
    float4 xn[MLR], dn[MLR];
    auto fetch = [&](int64_t p0) {
#pragma unroll
        for (int r = 0; r < MLR; r++) {
            xn[r] = z4; dn[r] = z4;                       // (rows past the end: zero statistics and zero dfeats -> zero contributions, no store)
            if (ok && p0 + r < p_end) {
                xn[r] = ld_stream_f4(x + (b * Tp + 1 + p0 + r) * D + 4 * t);
                dn[r] = ld_stream_f4(dfeats + (b * P + p0 + r) * D + 4 * t);
            }
        }
    };
    fetch(p_begin);
    for (int64_t p0 = p_begin; p0 < p_end; p0 += MLR) {
        float4 xh[MLR], zh[MLR], gz[MLR];
        float r1[MLR], r2[MLR];
        {
            float4 xv[MLR], df[MLR];
            float2 s1[MLR], s2[MLR];
#pragma unroll
            for (int r = 0; r < MLR; r++) {
                xv[r] = xn[r]; df[r] = dn[r];
                s1[r] = make_float2(0.f, 0.f); s2[r] = make_float2(0.f, 0.f);
                if (p0 + r < p_end) { s1[r] = stats1[b * Tp + 1 + p0 + r]; s2[r] = stats2[b * P + p0 + r]; }
            }
            if (p0 + MLR < p_end) fetch(p0 + MLR);
#pragma unroll
            for (int r = 0; r < MLR; r++) {
                r1[r] = s1[r].y; r2[r] = s2[r].y;
                xh[r] = make_float4((xv[r].x - s1[r].x) * s1[r].y, (xv[r].y - s1[r].x) * s1[r].y, (xv[r].z - s1[r].x) * s1[r].y, (xv[r].w - s1[r].x) * s1[r].y);
                const float4 y = make_float4(fmaf(xh[r].x, ga.x, be.x), fmaf(xh[r].y, ga.y, be.y), fmaf(xh[r].z, ga.z, be.z), fmaf(xh[r].w, ga.w, be.w));
                zh[r] = make_float4((y.x * cv.x - s2[r].x) * s2[r].y, (y.y * cv.y - s2[r].x) * s2[r].y, (y.z * cv.z - s2[r].x) * s2[r].y, (y.w * cv.w - s2[r].x) * s2[r].y);
                gz[r] = make_float4(df[r].x * gb.x, df[r].y * gb.y, df[r].z * gb.z, df[r].w * gb.w);
                a_g2.x += df[r].x * zh[r].x; a_g2.y += df[r].y * zh[r].y; a_g2.z += df[r].z * zh[r].z; a_g2.w += df[r].w * zh[r].w;
                a_b2.x += df[r].x; a_b2.y += df[r].y; a_b2.z += df[r].z; a_b2.w += df[r].w;
                const float m1 = wave_sum(gz[r].x + gz[r].y + gz[r].z + gz[r].w);
                const float m2 = wave_sum(gz[r].x * zh[r].x + gz[r].y * zh[r].y + gz[r].z * zh[r].z + gz[r].w * zh[r].w);
                if (lane == 0) redm[0][w][r] = make_float2(m1, m2);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MLR; r++) {
            const float2 q0 = redm[0][0][r], q1 = redm[0][1][r], q2 = redm[0][2][r], q3 = redm[0][3][r];
            const float m1 = (q0.x + q1.x + q2.x + q3.x) * invD, m2 = (q0.y + q1.y + q2.y + q3.y) * invD;
            const float4 y = make_float4(fmaf(xh[r].x, ga.x, be.x), fmaf(xh[r].y, ga.y, be.y), fmaf(xh[r].z, ga.z, be.z), fmaf(xh[r].w, ga.w, be.w));
            const float4 dz = make_float4(r2[r] * (gz[r].x - m1 - zh[r].x * m2), r2[r] * (gz[r].y - m1 - zh[r].y * m2),
                                          r2[r] * (gz[r].z - m1 - zh[r].z * m2), r2[r] * (gz[r].w - m1 - zh[r].w * m2));
            a_c.x += dz.x * y.x; a_c.y += dz.y * y.y; a_c.z += dz.z * y.z; a_c.w += dz.w * y.w;
            const float4 dy = make_float4(dz.x * cv.x, dz.y * cv.y, dz.z * cv.z, dz.w * cv.w);
            a_g1.x += dy.x * xh[r].x; a_g1.y += dy.y * xh[r].y; a_g1.z += dy.z * xh[r].z; a_g1.w += dy.w * xh[r].w;
            a_b1.x += dy.x; a_b1.y += dy.y; a_b1.z += dy.z; a_b1.w += dy.w;
            gz[r] = make_float4(dy.x * ga.x, dy.y * ga.y, dy.z * ga.z, dy.w * ga.w);               // (g.d from here on)
            const float n1 = wave_sum(gz[r].x + gz[r].y + gz[r].z + gz[r].w);
            const float n2 = wave_sum(gz[r].x * xh[r].x + gz[r].y * xh[r].y + gz[r].z * xh[r].z + gz[r].w * xh[r].w);
            if (lane == 0) redm[1][w][r] = make_float2(n1, n2);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MLR; r++) {
            if (!ok || p0 + r >= p_end) continue;
            const float2 q0 = redm[1][0][r], q1 = redm[1][1][r], q2 = redm[1][2][r], q3 = redm[1][3][r];
            const float n1 = (q0.x + q1.x + q2.x + q3.x) * invD, n2 = (q0.y + q1.y + q2.y + q3.y) * invD;
            const int64_t xrow = b * Tp + 1 + p0 + r;
            const float4 o = make_float4(r1[r] * (gz[r].x - n1 - xh[r].x * n2), r1[r] * (gz[r].y - n1 - xh[r].y * n2),
                                         r1[r] * (gz[r].z - n1 - xh[r].z * n2), r1[r] * (gz[r].w - n1 - xh[r].w * n2));
            st_stream_f4(dx + xrow * D + 4 * t, o);
            a_dx.x += o.x; a_dx.y += o.y; a_dx.z += o.z; a_dx.w += o.w;
            if (dx_bf16) {                              // the bf16 copy the first dX GEMM reads (saves a separate cast pass)
                uint2 ob; ob.x = pack_bf2(o.x, o.y); ob.y = pack_bf2(o.z, o.w);
                ((uint2*)(dx_bf16 + xrow * D))[t] = ob;
            }
        }
    }
    // this workgroup's slab part[b][blockIdx.x][{dcls, dg1, db1, dg2, db2, sum dx}][D]: every thread owns its columns
    if (ok) {
        float* mine = part + ((int64_t)b * gridDim.x + blockIdx.x) * 6 * D;
        ((float4*)mine)[t] = a_c; ((float4*)(mine + D))[t] = a_g1; ((float4*)(mine + 2 * D))[t] = a_b1;
        ((float4*)(mine + 3 * D))[t] = a_g2; ((float4*)(mine + 4 * D))[t] = a_b2; ((float4*)(mine + 5 * D))[t] = a_dx;
    }
}

// cls rows: dy0 = dcls[b,:] -> LN1 backward on token 0 of image b
__global__ __launch_bounds__(64) void cls_ln_bwd_kernel(const float* __restrict__ dcls, const float* __restrict__ x,
                                                        const float2* __restrict__ stats1, const float* __restrict__ g1, float* dx,
                                                        float* part, int64_t Tp, int D, bf16_t* dx_bf16) {
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, row = b * Tp;
    const int nvec = D >> 2;
    const float2 st = stats1[row];
    float4 xh[LN_MAXV], gd[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 xv = ((const float4*)(x + row * D))[idx], dy = ((const float4*)(dcls + b * D))[idx], g = ((const float4*)g1)[idx];
            xh[i] = make_float4((xv.x - st.x) * st.y, (xv.y - st.x) * st.y, (xv.z - st.x) * st.y, (xv.w - st.x) * st.y);
            gd[i] = make_float4(dy.x * g.x, dy.y * g.y, dy.z * g.z, dy.w * g.w);
            s1 += gd[i].x + gd[i].y + gd[i].z + gd[i].w;
            s2 += gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w;
            // this image's contribution to (dg1, db1) (and, below, its dx row): part[b][{dg1, db1, dx}][D], reduced in image order afterwards
            ((float4*)(part + b * 3 * D))[idx] = make_float4(dy.x * xh[i].x, dy.y * xh[i].y, dy.z * xh[i].z, dy.w * xh[i].w);
            ((float4*)(part + b * 3 * D + D))[idx] = dy;
        }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const float4 o = make_float4(st.y * (gd[i].x - s1 - xh[i].x * s2), st.y * (gd[i].y - s1 - xh[i].y * s2),
                                         st.y * (gd[i].z - s1 - xh[i].z * s2), st.y * (gd[i].w - s1 - xh[i].w * s2));
            ((float4*)(dx + row * D))[idx] = o;
            ((float4*)(part + b * 3 * D + 2 * D))[idx] = o;
            if (dx_bf16) {
                uint2 ob; ob.x = pack_bf2(o.x, o.y); ob.y = pack_bf2(o.z, o.w);
                ((uint2*)(dx_bf16 + row * D))[idx] = ob;
            }
        }
    }
}

OWL_API int owl_merge_ln_bwd(void* stream, const float* dfeats, const float* x, const float* cls_ln, const float* stats1,
                                const float* stats2, const float* g1, const float* b1, const float* g2, float* dx, float* dcls_ws,
                                float* dg1, float* db1, float* dg2, float* db2, int64_t B, int64_t P, int64_t Tp, int64_t D,
                                float* partials, int64_t partials_floats, void* dx_bf16, float* dx_colsum) {
    OWL_CHECK_ARG(dfeats && x && cls_ln && stats1 && stats2 && g1 && b1 && g2 && dx && dcls_ws && dg1 && db1 && dg2 && db2 && partials, "owl_merge_ln_bwd: null pointer");
    OWL_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV, "owl_merge_ln_bwd: D must be a multiple of 4 and <= 1024");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 64;
    const int nbx = (int)((P + rpb - 1) / rpb);
    OWL_CHECK_ARG(partials_floats >= B * nbx * 6 * D, "owl_merge_ln_bwd: needs %lld floats of partial-sum scratch (6 D per 64-row block)", (long long)(B * nbx * 6 * D));
    hipLaunchKernelGGL(merge_ln_bwd_kernel, dim3((unsigned)nbx, (unsigned)B), dim3(256), 0, s, dfeats, x, cls_ln,
                       (const float2*)stats1, (const float2*)stats2, g1, b1, g2, dx, partials, P, Tp, (int)D, rpb, (bf16_t*)dx_bf16);
    OWL_LAUNCH_CHECK();
    // d(cls_ln)[b] = sum over the image's row blocks (per-image groups); the four LN parameter gradients (and, on request, the column sums of
    // dx: the bias gradient of the linear layer that produced this residual position) += sum over all slabs
    ReduceOuts oc{}; oc.o[0] = dcls_ws;
    int rc = partials_reduce(s, partials, oc, 1, (int)D, 6 * D, nbx, (int)B, (int64_t)nbx * 6 * D, D, 0);
    if (rc) return rc;
    ReduceOuts op{}; op.o[0] = dg1; op.o[1] = db1; op.o[2] = dg2; op.o[3] = db2; op.o[4] = dx_colsum;
    rc = partials_reduce(s, partials + D, op, dx_colsum ? 5 : 4, (int)D, 6 * D, (int)(B * nbx), 1, 0, 0, 1);
    if (rc) return rc;
    // class-token rows: the slabs above have been consumed (stream order), so the scratch is reused for part[b][{dg1, db1, dx}][D]
    hipLaunchKernelGGL(cls_ln_bwd_kernel, dim3((unsigned)B), dim3(64), 0, s, dcls_ws, x, (const float2*)stats1, g1, dx, partials, Tp, (int)D, (bf16_t*)dx_bf16);
    OWL_LAUNCH_CHECK();
    ReduceOuts o2{}; o2.o[0] = dg1; o2.o[1] = db1; o2.o[2] = dx_colsum;
    return partials_reduce(s, partials, o2, dx_colsum ? 3 : 2, (int)D, 3 * D, (int)B, 1, 0, 0, 1);
}

// ---------------------------------------------------------------------------------------------------
// class head backward (ref src/models.py:25-36): s_c = inv * (e . qhat_j*), inv = 1/(|e|+1e-6)
//   de = inv * sum_c g_c qhat_{j*c} - (sum_c g_c s_c) * inv * e/|e| ;  dqhat_j += sum_rows [j = j*] g_c inv e
// One wave per row (row-parallel, no serial loop): writes de (bf16) and the routed, scaled upstream
// G[r, j] = g_c * inv_r * [j == 3c + argmax] (bf16 [rows, 32]); dqhat = G^T e is then an ordinary split-K GEMM.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void class_sims_bwd_kernel(const float* __restrict__ dsims, const float* __restrict__ sims,
                                                             const unsigned char* __restrict__ argmax, const float* __restrict__ inv_norm,
                                                             const float* __restrict__ e, const float* __restrict__ qhat, bf16_t* de,
                                                             bf16_t* G, bf16_t* e_bf16, int64_t rows, int Dt, int C, int rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lq[];     // qhat [32][Dt]
    for (int i = threadIdx.x; i < 32 * (Dt >> 2); i += 512) ((float4*)lq)[i] = ((const float4*)qhat)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the 64 KiB query table is loaded once per workgroup of 8 waves (2 workgroups = 16 waves per CU: the per-row chain
    // load -> reduce -> load -> store is latency-bound, so occupancy matters): many rows per wave at large batch, fewer when
    // that would leave most CUs idle (batch 1).  A wave handles its rows two at a time (two independent chains).
    const int64_t row0 = ((int64_t)blockIdx.x * 8 + w) * rows_per_wave;
    const int64_t row1 = min(rows, row0 + rows_per_wave);
    auto head = [&](int64_t r, int& jsel, float& gcls, float& coef_e) {
        const float inv = inv_norm[r];
        // lane c < C owns class c of this row: its arg-max prompt and upstream gradient are read ONCE and handed to the
        // feature loop through v_readlane (wave-uniform there: scalar operands, no shuffles, no byte loads per class)
        jsel = 0; gcls = 0.f;
        float gs = 0.f;
        if (lane < C) {
            const float ds = dsims[r * C + lane];
            jsel = 3 * lane + (int)argmax[r * C + lane];
            gcls = ds * inv;
            gs = ds * sims[r * C + lane];
        }
        gs = wave_sum(gs);
        // routed upstream G[r][j] = g_c * inv * [j == 3c + argmax_c]: lane j looks its class up
        const int c = lane / 3;
        const int jc = __shfl(jsel, c < C ? c : 0, 64);
        const float gc = __shfl(gcls, c < C ? c : 0, 64);
        if (lane < 32) G[r * 32 + lane] = f2bf((c < C && jc == lane) ? gc : 0.f);
        const float nrm = 1.0f / inv - 1e-6f;
        coef_e = gs * inv / nrm;
    };
    auto body = [&](int64_t r, int jsel, float gcls, float coef_e) {
        for (int k4 = lane; k4 < (Dt >> 2); k4 += 64) {
            const float4 ev = ((const float4*)(e + r * Dt))[k4];
            float4 d = make_float4(-coef_e * ev.x, -coef_e * ev.y, -coef_e * ev.z, -coef_e * ev.w);
            for (int c = 0; c < C; c++) {
                const int j = __builtin_amdgcn_readlane(jsel, c);
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gcls), c));
                const float4 q = ((const float4*)(lq + j * Dt))[k4];
                d.x += g * q.x; d.y += g * q.y; d.z += g * q.z; d.w += g * q.w;
            }
            uint2 o; o.x = pack_bf2(d.x, d.y); o.y = pack_bf2(d.z, d.w);
            ((uint2*)(de + r * Dt))[k4] = o;
            uint2 eb; eb.x = pack_bf2(ev.x, ev.y); eb.y = pack_bf2(ev.z, ev.w);
            ((uint2*)(e_bf16 + r * Dt))[k4] = eb;
        }
    };
    int64_t r = row0;
    for (; r + 1 < row1; r += 2) {
        int ja, jb; float ga, gb, ca, cb;
        head(r, ja, ga, ca);
        head(r + 1, jb, gb, cb);
        body(r, ja, ga, ca);
        body(r + 1, jb, gb, cb);
    }
    if (r < row1) {
        int ja; float ga, ca;
        head(r, ja, ga, ca);
        body(r, ja, ga, ca);
    }
}

// The same kernel for Dt = 256 NI (the model widths: 512, 768), rows software-pipelined: the per-row chain above (head loads -> wave reduction -> row
// loads -> stores) leaves the memory system idle between its two load phases; here every load a row needs (its e row: NI float4 per lane; the
// class gradients, arg-max prompts and sims of lanes < C; the saved 1/norm) is issued THREE rows ahead into one of four register sets used in
// rotation.  The arithmetic per row and its order are those of class_sims_bwd_kernel: same bits (tests/test_kernels_gpu.py compares the two).
template <int NI>
__global__ __launch_bounds__(512) void class_sims_bwd_pf_kernel(const float* __restrict__ dsims, const float* __restrict__ sims,
                                                                const unsigned char* __restrict__ argmax, const float* __restrict__ inv_norm,
                                                                const float* __restrict__ e, const float* __restrict__ qhat, bf16_t* de,
                                                                bf16_t* G, bf16_t* e_bf16, int64_t rows, int C, int rows_per_wave) {
    constexpr int Dt = 256 * NI;
    extern __shared__ __attribute__((aligned(16))) float lq[];     // qhat [32][Dt]
    for (int i = threadIdx.x; i < 32 * (Dt >> 2); i += 512) ((float4*)lq)[i] = ((const float4*)qhat)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 8 + w) * rows_per_wave;
    const int64_t row1 = min(rows, row0 + rows_per_wave);
    if (row0 >= row1) return;
    struct RowIn { float4 ev[NI]; float ds, sm, inv; int am; };
    auto fetch = [&](int64_t r, RowIn& x) {
        r = min(r, row1 - 1);                                      // unconditional (a branch around loads costs a vmcnt(0)); the surplus fetches are dropped
#pragma unroll
        for (int i = 0; i < NI; i++) x.ev[i] = ((const float4*)(e + r * Dt))[lane + 64 * i];
        x.inv = inv_norm[r];
        const int64_t ci = r * C + (lane < C ? lane : 0);
        x.ds = dsims[ci]; x.am = (int)argmax[ci]; x.sm = sims[ci];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto process = [&](int64_t r, const RowIn& x) {
        const float inv = x.inv;
        int jsel = 0; float gcls = 0.f, gs = 0.f;
        if (lane < C) {
            jsel = 3 * lane + x.am;
            gcls = x.ds * inv;
            gs = x.ds * x.sm;
        }
        gs = wave_sum(gs);
        const int c = lane / 3;
        const int jc = __shfl(jsel, c < C ? c : 0, 64);
        const float gc = __shfl(gcls, c < C ? c : 0, 64);
        if (lane < 32) G[r * 32 + lane] = f2bf((c < C && jc == lane) ? gc : 0.f);
        const float nrm = 1.0f / inv - 1e-6f;
        const float coef_e = gs * inv / nrm;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int k4 = lane + 64 * i;
            const float4 ev = x.ev[i];
            float4 d = make_float4(-coef_e * ev.x, -coef_e * ev.y, -coef_e * ev.z, -coef_e * ev.w);
            for (int cc = 0; cc < C; cc++) {
                const int j = __builtin_amdgcn_readlane(jsel, cc);
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gcls), cc));
                const float4 q = ((const float4*)(lq + j * Dt))[k4];
                d.x += g * q.x; d.y += g * q.y; d.z += g * q.z; d.w += g * q.w;
            }
            uint2 o; o.x = pack_bf2(d.x, d.y); o.y = pack_bf2(d.z, d.w);
            ((uint2*)(de + r * Dt))[k4] = o;
            uint2 eb; eb.x = pack_bf2(ev.x, ev.y); eb.y = pack_bf2(ev.z, ev.w);
            ((uint2*)(e_bf16 + r * Dt))[k4] = eb;
        }
    };
    RowIn x0, x1, x2, x3;
    fetch(row0, x0); fetch(row0 + 1, x1); fetch(row0 + 2, x2);
    for (int64_t r = row0; r < row1; r += 4) {
        fetch(r + 3, x3); process(r, x0);
        if (r + 1 >= row1) break;
        fetch(r + 4, x0); process(r + 1, x1);
        if (r + 2 >= row1) break;
        fetch(r + 5, x1); process(r + 2, x2);
        if (r + 3 >= row1) break;
        fetch(r + 6, x2); process(r + 3, x3);
    }
}

// dQ from dqhat: qhat = Q/|Q| + 1e-6  ->  dQ = (dqhat - (dqhat . Qn) Qn) / |Q|,  Qn = Q/|Q|
__global__ __launch_bounds__(64) void qhat_bwd_kernel(const float* __restrict__ dqhat, const float* __restrict__ q, float* dq, int Dt) {
    const int j = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f, dt = 0.f;
    for (int k = lane; k < Dt; k += 64) { const float v = q[(int64_t)j * Dt + k]; ss += v * v; dt += v * dqhat[(int64_t)j * Dt + k]; }
    ss = wave_sum(ss); dt = wave_sum(dt);
    const float n = sqrtf(ss);
    for (int k = lane; k < Dt; k += 64) {
        const float qn = q[(int64_t)j * Dt + k] / n;
        dq[(int64_t)j * Dt + k] += (dqhat[(int64_t)j * Dt + k] - (dt / n) * qn) / n;
    }
}

OWL_API int owl_class_sims_bwd(void* stream, const float* dsims, const float* sims, const unsigned char* argmax, const float* inv_norm,
                                  const float* e, const float* qhat32, void* de_bf16, void* g_bf16, void* e_bf16, int64_t rows,
                                  int64_t Dt, int64_t C) {
    OWL_CHECK_ARG(dsims && sims && argmax && inv_norm && e && qhat32 && de_bf16 && g_bf16 && e_bf16, "owl_class_sims_bwd: null pointer");
    OWL_CHECK_ARG(3 * C <= 32 && Dt % 4 == 0, "owl_class_sims_bwd: 3*C <= 32, Dt %% 4");
    const size_t shmem = (size_t)32 * Dt * sizeof(float);
    OWL_CHECK_ARG(shmem <= 150 * 1024, "owl_class_sims_bwd: Dt too large for LDS");
    static unsigned long long attr_done = 0;
    OWL_ONCE_PER_DEVICE(attr_done, (void)hipFuncSetAttribute((const void*)class_sims_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    int rpw = (int)((rows + 8 * 512 - 1) / (8 * 512));           // aim at >= 512 workgroups ...
    rpw = rpw < 2 ? 2 : (rpw > 18 ? 18 : rpw);                   // ... with 16..144 rows each
    const dim3 grid((unsigned)((rows + 8 * rpw - 1) / (8 * rpw)));
    if (Dt == 512 || Dt == 768) {                                // the model widths: rows software-pipelined (same bits)
        static unsigned long long attr_pf = 0;
        OWL_ONCE_PER_DEVICE(attr_pf, ((void)hipFuncSetAttribute((const void*)class_sims_bwd_pf_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
                                      (void)hipFuncSetAttribute((const void*)class_sims_bwd_pf_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)));
        if (Dt == 512)
            hipLaunchKernelGGL(class_sims_bwd_pf_kernel<2>, grid, dim3(512), shmem, (hipStream_t)stream, dsims, sims, argmax, inv_norm, e, qhat32,
                               (bf16_t*)de_bf16, (bf16_t*)g_bf16, (bf16_t*)e_bf16, rows, (int)C, rpw);
        else
            hipLaunchKernelGGL(class_sims_bwd_pf_kernel<3>, grid, dim3(512), shmem, (hipStream_t)stream, dsims, sims, argmax, inv_norm, e, qhat32,
                               (bf16_t*)de_bf16, (bf16_t*)g_bf16, (bf16_t*)e_bf16, rows, (int)C, rpw);
        OWL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(class_sims_bwd_kernel, grid, dim3(512), shmem, (hipStream_t)stream, dsims, sims, argmax,
                       inv_norm, e, qhat32, (bf16_t*)de_bf16, (bf16_t*)g_bf16, (bf16_t*)e_bf16, rows, (int)Dt, (int)C, rpw);
    OWL_LAUNCH_CHECK();
    return 0;
}

// dqueries += d(qhat -> Q) of dqhat (f32 [32, Dt], rows < nq used)
OWL_API int owl_query_normalize_bwd(void* stream, const float* dqhat, const float* queries, float* dqueries, int64_t nq, int64_t Dt) {
    OWL_CHECK_ARG(dqhat && queries && dqueries && nq >= 1 && nq <= 32, "owl_query_normalize_bwd: bad args");
    hipLaunchKernelGGL(qhat_bwd_kernel, dim3((unsigned)nq), dim3(64), 0, (hipStream_t)stream, dqhat, queries, dqueries, (int)Dt);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// box head tail backward: d(xyxy) -> d(cx,cy,w,h) -> sigmoid' -> dense2 backward fused with dense1's
// erf-GELU derivative.  du1 = (dpre . W2) * gelu'(u1) (bf16) ; dW2 += dpre^T h1 ; db2 += sum dpre.
// ---------------------------------------------------------------------------------------------------

// One thread owns FOUR consecutive columns (8-byte loads of h1 / u1, 8-byte store of du1; its 16 dense2 weights live in registers) and walks
// the workgroup's rows four at a time (eight loads in flight); D <= 1024, D % 4 == 0.  Per column the arithmetic and the row order of the
// dW2 sums are those of a plain row loop.
__global__ __launch_bounds__(256) void box_final_bwd_kernel(const float* __restrict__ dboxes, const float* __restrict__ sig,
                                                            const bf16_t* __restrict__ h1, const bf16_t* __restrict__ u1,
                                                            const float* __restrict__ w2, bf16_t* __restrict__ du1, float* __restrict__ part,
                                                            int64_t rows, int D, int rows_per_block) {
    __shared__ float4 dpre_s[256];
    const int t = threadIdx.x;
    const int col = 4 * t;
    const bool col_ok = col < D;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    float accw[4][4];   // [k][column]
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 4; c++) accw[k][c] = 0.f;
    float4 wk[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (col_ok) {
#pragma unroll
        for (int k = 0; k < 4; k++) wk[k] = *(const float4*)(w2 + (int64_t)k * D + col);
    }
    float accu[4] = {0.f, 0.f, 0.f, 0.f};     // column sums of du1 (f32, before the bf16 rounding): dense1's bias gradient
    float4 accb = make_float4(0, 0, 0, 0);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 256) {
        const int64_t r = r0 + t;
        float4 dp = make_float4(0, 0, 0, 0);
        if (r < r_end) {
            const float4 g = *(const float4*)(dboxes + r * 4), s = *(const float4*)(sig + r * 4);
            const float dcx = g.x + g.z, dcy = g.y + g.w, dw = 0.5f * (g.z - g.x), dh = 0.5f * (g.w - g.y);
            dp = make_float4(dcx * s.x * (1.f - s.x), dcy * s.y * (1.f - s.y), dw * s.z * (1.f - s.z), dh * s.w * (1.f - s.w));
            accb.x += dp.x; accb.y += dp.y; accb.z += dp.z; accb.w += dp.w;
        }
        __syncthreads();
        dpre_s[t] = dp;
        __syncthreads();
        const int nr = (int)min((int64_t)256, r_end - r0);
        if (col_ok) {
            uint2 hn[4], un[4];                          // the NEXT four rows' operands: in flight while these four are computed
            auto fetch = [&](int rr) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int64_t row = r0 + min(rr + j, nr - 1);
                    hn[j] = ld_stream_u2(h1 + row * D + col);
                    un[j] = ld_stream_u2(u1 + row * D + col);
                }
            };
            fetch(0);
            for (int rr = 0; rr < nr; rr += 4) {
                uint2 hq[4], uq[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { hq[j] = hn[j]; uq[j] = un[j]; }
                if (rr + 4 < nr) fetch(rr + 4);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (rr + j >= nr) break;
                    const float4 d = dpre_s[rr + j];
                    const float hv[4] = {bf2f(hq[j].x & 0xffff), bf2f(hq[j].x >> 16), bf2f(hq[j].y & 0xffff), bf2f(hq[j].y >> 16)};
                    const float uv[4] = {bf2f(uq[j].x & 0xffff), bf2f(uq[j].x >> 16), bf2f(uq[j].y & 0xffff), bf2f(uq[j].y >> 16)};
                    const float w0[4] = {wk[0].x, wk[0].y, wk[0].z, wk[0].w}, w1[4] = {wk[1].x, wk[1].y, wk[1].z, wk[1].w};
                    const float w2r[4] = {wk[2].x, wk[2].y, wk[2].z, wk[2].w}, w3[4] = {wk[3].x, wk[3].y, wk[3].z, wk[3].w};
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float dh1 = d.x * w0[c] + d.y * w1[c] + d.z * w2r[c] + d.w * w3[c];
                        o[c] = dh1 * dgelu_erf_f(uv[c]);
                        accu[c] += o[c];
                        accw[0][c] += d.x * hv[c]; accw[1][c] += d.y * hv[c]; accw[2][c] += d.z * hv[c]; accw[3][c] += d.w * hv[c];
                    }
                    uint2 ov; ov.x = pack_bf2(o[0], o[1]); ov.y = pack_bf2(o[2], o[3]);
                    *(uint2*)(du1 + (r0 + rr + j) * D + col) = ov;
                }
            }
        }
    }
    // per-workgroup partials [nblk][{dW2 (4 D), db2 (4), sum du1 (D)}] (reduced deterministically by owl_slab_reduce)
    float* mypart = part + (int64_t)blockIdx.x * (5 * D + 4);
    if (col_ok) {
#pragma unroll
        for (int k = 0; k < 4; k++) *(float4*)(mypart + (int64_t)k * D + col) = make_float4(accw[k][0], accw[k][1], accw[k][2], accw[k][3]);
        *(float4*)(mypart + 4 * D + 4 + col) = make_float4(accu[0], accu[1], accu[2], accu[3]);
    }
    __shared__ float4 redb[4];
    accb.x = wave_sum(accb.x); accb.y = wave_sum(accb.y); accb.z = wave_sum(accb.z); accb.w = wave_sum(accb.w);
    if ((t & 63) == 0) redb[t >> 6] = accb;
    __syncthreads();
    if (t == 0) {
        const float4 a0 = redb[0], a1 = redb[1], a2 = redb[2], a3 = redb[3];
        *(float4*)(mypart + 4 * D) = make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
    }
}

// partials: f32 workspace [owl_box_final_bwd_blocks(rows)][5*D + 4]; dw2 [4,D] and db2 [4] must be CONTIGUOUS
// (dw2 followed by db2, as in the flat gradient bucket) -- they are accumulated by one deterministic reduce.  du1_colsum (optional, [D]):
// += column sums of du1 (dense1's bias gradient) from the same pass.
// rows per workgroup: 64 at large batch, down to 8 when that would leave most CUs idle (batch 1: 2304 rows)
static int box_bwd_rpb(int64_t rows) {
    int64_t r = (rows + 511) / 512;
    return (int)(r < 8 ? 8 : (r > 64 ? 64 : r));
}
OWL_API int owl_box_final_bwd_blocks(int64_t rows) { const int rpb = box_bwd_rpb(rows); return (int)((rows + rpb - 1) / rpb); }

OWL_API int owl_box_final_bwd(void* stream, const float* dboxes, const float* sig, const void* h1_bf16, const void* u1_bf16,
                                 const float* w2, void* du1_bf16, float* partials, float* dw2_db2, int64_t rows, int64_t D, float* du1_colsum) {
    OWL_CHECK_ARG(dboxes && sig && h1_bf16 && u1_bf16 && w2 && du1_bf16 && partials && dw2_db2, "owl_box_final_bwd: null pointer");
    OWL_CHECK_ARG(D <= 1024 && D % 4 == 0, "owl_box_final_bwd: D <= 1024, D %% 4");
    const int rpb = box_bwd_rpb(rows);
    const int nblk = (int)((rows + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(box_final_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dboxes, sig, (const bf16_t*)h1_bf16,
                       (const bf16_t*)u1_bf16, w2, (bf16_t*)du1_bf16, partials, rows, (int)D, rpb);
    OWL_LAUNCH_CHECK();
    int rc = owl_slab_reduce_impl(s, partials, dw2_db2, 4 * D + 4, 5 * D + 4, nblk, 1);
    if (rc || !du1_colsum) return rc;
    return owl_slab_reduce_impl(s, partials + 4 * D + 4, du1_colsum, D, 5 * D + 4, nblk, 1);
}

// ---------------------------------------------------------------------------------------------------
// transpose bf16 [R,C] -> [C,R] with optional column sums (bias gradient) in the same pass.
// ---------------------------------------------------------------------------------------------------
// 64x64 tiles, 16-byte global accesses on both sides: a thread loads 8 consecutive columns of a row (uint4), the tile
// goes through LDS (row stride 72 halves = 144 B keeps 16-byte alignment and spreads banks), and is written back as 8
// consecutive rows (= 8 consecutive output columns) per thread.  R must be a multiple of 8 for the vector store path
// (token counts are; the scalar tail handles the rest), C a multiple of 8.
__global__ __launch_bounds__(256) void transpose_colsum_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                               int64_t ld_out, float* colsum, int64_t R, int64_t C, int row_tiles, int64_t Cpad) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
    __shared__ float cs[32][65];
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int lr = t >> 3, lc = (t & 7) * 8;          // load: rows lr, lr+32 ; columns lc..lc+7
    const int sc = t >> 3, sr = (t & 7) * 8;          // store: output rows (= input cols) sc, sc+32 ; input rows sr..sr+7
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int rt = 0; rt < row_tiles; rt++) {
        const int64_t r0 = ((int64_t)blockIdx.y * row_tiles + rt) * 64;
        if (r0 >= R) break;
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t r = r0 + lr + 32 * h, c = c0 + lc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r < R && c + 8 <= C) v = *(const uint4*)(in + r * ld_in + c);
            else if (r < R) { bf16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; for (int e = 0; e < 8; e++) if (c + e < C) tmp[e] = in[r * ld_in + c + e]; v = *(uint4*)tmp; }
            *(uint4*)&tile[lr + 32 * h][lc] = v;
            if (colsum) {
                const bf16_t* pv = (const bf16_t*)&v;
#pragma unroll
                for (int e = 0; e < 8; e++) acc[e] += bf2f(pv[e]);
            }
        }
        __syncthreads();
        if (out) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int col = sc + 32 * h;                 // input column = output row
                const int64_t oc = c0 + col, orow0 = r0 + sr;
                if (oc >= C) continue;
                bf16_t tmp[8];
#pragma unroll
                for (int e = 0; e < 8; e++) tmp[e] = tile[sr + e][col];
                if (orow0 + 8 <= R) *(uint4*)(out + oc * ld_out + orow0) = *(uint4*)tmp;
                else for (int e = 0; e < 8; e++) if (orow0 + e < R) out[oc * ld_out + orow0 + e] = tmp[e];
            }
        }
    }
    if (colsum) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; e++) cs[lr][lc + e] = acc[e];
        __syncthreads();
        if (t < 64 && c0 + t < C) {
            float s = 0.f;
            for (int k = 0; k < 32; k++) s += cs[k][t];
            colsum[(int64_t)blockIdx.y * Cpad + c0 + t] = s;         // this row-block's partial (colsum = the partials scratch here)
        }
    }
}

OWL_API int owl_transpose_colsum_bf16(void* stream, const void* in, int64_t ld_in, void* out_t, int64_t ld_out, float* colsum,
                                         int64_t R, int64_t C, float* partials, int64_t partials_floats) {
    OWL_CHECK_ARG(in && (out_t || colsum) && R > 0 && C > 0, "owl_transpose_colsum_bf16: bad args");
    OWL_CHECK_ARG(ld_in % 8 == 0 && (!out_t || ld_out % 8 == 0), "owl_transpose_colsum_bf16: leading dimensions must be multiples of 8");
    const int row_tiles = 8;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 64 * row_tiles - 1) / (64 * row_tiles)));
    const int64_t Cpad = (C + 3) / 4 * 4;
    if (colsum) OWL_CHECK_ARG(C % 4 == 0 && partials && partials_floats >= (int64_t)grid.y * Cpad, "owl_transpose_colsum_bf16: column sums need C %% 4 == 0 and %lld floats of partial-sum scratch", (long long)grid.y * Cpad);
    hipLaunchKernelGGL(transpose_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out_t, ld_out,
                       colsum ? partials : nullptr, R, C, row_tiles, Cpad);
    OWL_LAUNCH_CHECK();
    if (!colsum) return 0;
    ReduceOuts o{}; o.o[0] = colsum;
    return partials_reduce((hipStream_t)stream, partials, o, 1, (int)C, Cpad, (int)grid.y, 1, 0, 0, 1);
}

// f32 column sums (bias gradient of an f32 upstream, e.g. the residual-stream gradient); colsum += sum_r in[r][c].
// A workgroup owns 256 columns x 256 rows: lane -> 4 columns (16-byte loads), its 4 waves take rows r0+w, r0+w+4, ...;
// partials meet in LDS, one partial sum per column and workgroup, reduced in fixed order afterwards (the first version: one thread per column, 1.9 TB/s).
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, float* colsum, int64_t R, int64_t C) {
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 256 + lane * 4;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(R, r0 + 256);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const float4 v = *(const float4*)(in + r * C + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[w][lane * 4 + 0] = acc.x; part[w][lane * 4 + 1] = acc.y; part[w][lane * 4 + 2] = acc.z; part[w][lane * 4 + 3] = acc.w;
    __syncthreads();
    const int64_t cc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cc < C) colsum[(int64_t)blockIdx.y * C + cc] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

OWL_API int owl_colsum_f32(void* stream, const float* in, float* colsum, int64_t R, int64_t C, float* partials, int64_t partials_floats) {
    OWL_CHECK_ARG(in && colsum && partials && R > 0 && C > 0 && C % 4 == 0, "owl_colsum_f32: bad arguments (C %% 4 == 0)");
    const int gy = (int)((R + 255) / 256);
    OWL_CHECK_ARG(partials_floats >= (int64_t)gy * C, "owl_colsum_f32: needs %lld floats of partial-sum scratch", (long long)gy * C);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, in, partials, R, C);
    OWL_LAUNCH_CHECK();
    ReduceOuts o{}; o.o[0] = colsum;
    return partials_reduce((hipStream_t)stream, partials, o, 1, (int)C, C, gy, 1, 0, 0, 1);
}

// bf16 column sums (bias gradient when the weight gradient reads dY in place: gemm_tn.hip); colsum += sum_r in[r][c].
// A workgroup owns 512 columns x 256 rows: lane -> 8 columns (16-byte loads, a wave reads 1 KiB of a row), its 4 waves
// take rows r0+w, r0+w+4, ...; partial sums meet in LDS, one partial sum per column and workgroup (reduced in fixed order afterwards).  HBM-bound:
// reads R*C*2 bytes once.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ in, int64_t ld, float* colsum, int64_t R, int64_t C) {
    __shared__ float part[4][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 512 + lane * 8;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(R, r0 + 256);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const uint4 u = *(const uint4*)(in + r * ld + c);
            const unsigned wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; e++) { acc[2 * e] += __uint_as_float(wds[e] << 16); acc[2 * e + 1] += __uint_as_float(wds[e] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) part[w][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int64_t cc = (int64_t)blockIdx.x * 512 + i;
        if (cc < C) colsum[(int64_t)blockIdx.y * C + cc] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    }
}

OWL_API int owl_colsum_bf16(void* stream, const void* in_bf16, int64_t ld, float* colsum, int64_t R, int64_t C, float* partials, int64_t partials_floats) {
    OWL_CHECK_ARG(in_bf16 && colsum && partials && R > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0, "owl_colsum_bf16: bad arguments (C, ld %% 8 == 0)");
    const int gy = (int)((R + 255) / 256);
    OWL_CHECK_ARG(partials_floats >= (int64_t)gy * C, "owl_colsum_bf16: needs %lld floats of partial-sum scratch", (long long)gy * C);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)((C + 511) / 512), (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in_bf16, ld, partials, R, C);
    OWL_LAUNCH_CHECK();
    ReduceOuts o{}; o.o[0] = colsum;
    return partials_reduce((hipStream_t)stream, partials, o, 1, (int)C, C, gy, 1, 0, 0, 1);
}
