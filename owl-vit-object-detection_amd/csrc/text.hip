// Query-bank initialisation: ref src/models.py:155-169 runs HF's CLIP-style text tower ONCE
// on 3 prompts per class and keeps `text_embeds` as the learnable query bank.  The tower's Linear / LayerNorm work reuses
// gemm.hip / norm.hip; this file holds the three pieces that have no counterpart on the vision path:
//   text_embed_kernel     token_embedding[ids] + position_embedding[t]                       (HF5:356-373)
//   causal_attn_kernel    softmax(QK^T/8 + causal mask) V for short sequences (S <= 64, dh=64) (HF5:377-402, 634-648)
//   text_pool_kernel      final LayerNorm of the EOS row (arg-max token id, HF5:651-657) -> text_projection (no bias,
//                         HF5:826,952) -> L2 normalise (HF5:958,970), all f32
// One-shot, latency-bound work (30 prompts x 16 tokens); no MFMA, no tuning beyond coalesced access.
#include "common.h"

__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* __restrict__ x, int S, int W,
                                                         int vocab) {
    const int m = blockIdx.x;
    int64_t id = ids[m];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* t = tok + id * W;
    const float* p = pos + (int64_t)(m % S) * W;
    for (int c = threadIdx.x; c < W; c += 256) x[(int64_t)m * W + c] = t[c] + p[c];
}

// One wave per (sequence, head); lane i = query i (< S <= 64).  K and V of the head are staged in LDS as f32.
__global__ __launch_bounds__(64) void causal_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S, int W,
                                                         float scale) {
    __shared__ float ks[64][65];
    __shared__ float vs[64][65];
    const int n = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* base = qkv + (int64_t)n * S * ld + h * 64;
    for (int e = lane; e < S * 64; e += 64) {
        const int j = e >> 6, d = e & 63;
        ks[j][d] = bf2f(base[j * ld + W + d]);
        vs[j][d] = bf2f(base[j * ld + 2 * W + d]);
    }
    __syncthreads();
    if (lane >= S) return;
    float q[64], acc[64];
#pragma unroll
    for (int d = 0; d < 64; d++) { q[d] = bf2f(base[lane * ld + d]) * scale; acc[d] = 0.f; }
    float mx = -INFINITY, l = 0.f;
    for (int j = 0; j <= lane; j++) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; d++) s += q[d] * ks[j][d];
        const float nm = fmaxf(mx, s);
        const float corr = __expf(mx - nm), p = __expf(s - nm);
        l = l * corr + p;
#pragma unroll
        for (int d = 0; d < 64; d++) acc[d] = acc[d] * corr + p * vs[j][d];
        mx = nm;
    }
    const float inv = 1.f / l;
    bf16_t* o = out + ((int64_t)n * S + lane) * W + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d++) o[d] = f2bf(acc[d] * inv);
}

// One 256-thread workgroup per sequence.
__global__ __launch_bounds__(256) void text_pool_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ wproj, float* __restrict__ out, int S, int W,
                                                        int Pdim, float eps) {
    extern __shared__ float sm[];          // y[W] | o[Pdim] | red[8]
    float* y = sm;
    float* o = sm + W;
    float* red = o + Pdim;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int eos = 0;
    {
        int64_t best = ids[(int64_t)n * S];
        for (int t = 1; t < S; t++) {
            const int64_t v = ids[(int64_t)n * S + t];
            if (v > best) { best = v; eos = t; }        // first maximal id, like torch.argmax
        }
    }
    const float* row = x + ((int64_t)n * S + eos) * W;
    float s = 0.f;
    for (int c = tid; c < W; c += 256) s += row[c];
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / W;
    __syncthreads();
    float v = 0.f;
    for (int c = tid; c < W; c += 256) { const float d = row[c] - mean; v += d * d; }
    v = wave_sum(v);
    if (lane == 0) red[w] = v;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / W + eps);
    for (int c = tid; c < W; c += 256) y[c] = (row[c] - mean) * rstd * gamma[c] + beta[c];
    __syncthreads();
    for (int p = w; p < Pdim; p += 4) {
        const float* wr = wproj + (int64_t)p * W;
        float a = 0.f;
        for (int c = lane; c < W; c += 64) a += y[c] * wr[c];
        a = wave_sum(a);
        if (lane == 0) o[p] = a;
    }
    __syncthreads();
    float q = 0.f;
    for (int p = tid; p < Pdim; p += 256) q += o[p] * o[p];
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[4 + w] = q;
    __syncthreads();
    const float nrm = sqrtf(red[4] + red[5] + red[6] + red[7]);
    for (int p = tid; p < Pdim; p += 256) out[(int64_t)n * Pdim + p] = o[p] / nrm;
}

OWL_API int owl_text_embed(void* stream, const int64_t* ids, const float* tok_emb, const float* pos_emb, float* x, int64_t N,
                              int64_t S, int64_t W, int64_t vocab) {
    OWL_CHECK_ARG(ids && tok_emb && pos_emb && x && N > 0 && S > 0 && W > 0 && vocab > 0, "owl_text_embed: bad arguments");
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)(N * S)), dim3(256), 0, (hipStream_t)stream, ids, tok_emb, pos_emb, x, (int)S,
                       (int)W, (int)vocab);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_causal_attention_small(void* stream, const void* qkv_bf16, void* out_bf16, int64_t N, int64_t S, int64_t heads,
                                          float scale) {
    OWL_CHECK_ARG(qkv_bf16 && out_bf16 && N > 0 && heads > 0, "owl_causal_attention_small: bad arguments");
    OWL_CHECK_ARG(S > 0 && S <= 64, "owl_causal_attention_small: sequence length %lld not in 1..64", (long long)S);
    hipLaunchKernelGGL(causal_attn_kernel, dim3((unsigned)N, (unsigned)heads), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv_bf16,
                       (bf16_t*)out_bf16, (int)S, (int)(heads * 64), scale);
    OWL_LAUNCH_CHECK();
    return 0;
}

OWL_API int owl_text_pool_project(void* stream, const float* x, const int64_t* ids, const float* gamma, const float* beta,
                                     const float* wproj, float* out, int64_t N, int64_t S, int64_t W, int64_t Pdim, float eps) {
    OWL_CHECK_ARG(x && ids && gamma && beta && wproj && out && N > 0 && S > 0 && W > 0 && Pdim > 0, "owl_text_pool_project: bad arguments");
    const size_t lds = (size_t)(W + Pdim + 8) * sizeof(float);
    OWL_CHECK_ARG(lds <= 64 * 1024, "owl_text_pool_project: W + Pdim too large for LDS");
    hipLaunchKernelGGL(text_pool_kernel, dim3((unsigned)N), dim3(256), lds, (hipStream_t)stream, x, ids, gamma, beta, wproj, out, (int)S,
                       (int)W, (int)Pdim, eps);
    OWL_LAUNCH_CHECK();
    return 0;
}
