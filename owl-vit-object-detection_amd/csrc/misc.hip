// Small HBM-bound utilities: f32 -> bf16 cast, bf16 2-D transpose, fills.
#include "common.h"

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const float4 a = *(const float4*)(in + i), b = *(const float4*)(in + i + 4);
            uint4 o;
            o.x = pack_bf2(a.x, a.y); o.y = pack_bf2(a.z, a.w); o.z = pack_bf2(b.x, b.y); o.w = pack_bf2(b.z, b.w);
            *(uint4*)(out + i) = o;
        } else {
            for (int64_t j = i; j < n; j++) out[j] = f2bf(in[j]);
        }
    }
}

extern "C" int owl_cast_f32_bf16(void* stream, const float* in, void* out, int64_t n) {
    OWL_CHECK_ARG(in && out && n >= 0, "owl_cast_f32_bf16: null pointer");
    OWL_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "owl_cast_f32_bf16: pointers must be 16-byte aligned");
    if (n == 0) return 0;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, n);
    OWL_LAUNCH_CHECK();
    return 0;
}

// out[c][r] = in[r][c]; in [R, ld_in] (R x C used), out [C, ld_out].  64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                        int64_t ld_out, int64_t R, int64_t C) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[c * ld_out + r] = tile[tx][i];
    }
}

// Per-image token transpose: out[b][c][t] = in[b*Tp + t][c] for c < ncols (the attention-backward operands Q^T / K^T / V^T
// from the row-major QKV the forward GEMM has just written: HBM-bound, 4 B per element, instead of a second GEMM with a
// transposing epilogue -- same bits, since both round the same f32 accumulator + bias once).  64 x 64 tiles; 16-byte global
// loads (8 columns of a token) are scattered into a [c][t] LDS image with a 66-element pitch, read back as dwords and stored
// as 16 bytes = 8 tokens of one column, 8 lanes per 128-byte output row segment.
template <int TT, int TC>      // tile: TT tokens x TC columns
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                               int Tp, int ncols) {   // ncols here = rows per image of `out`
    __shared__ bf16_t tile[TC][TT + 2];
    const int b = blockIdx.z, t0 = blockIdx.y * TT, c0 = blockIdx.x * TC;
    const int i = threadIdx.x;
    constexpr int CPR = TC / 8;                      // 16-byte chunks per token row of the tile
    constexpr int RPI = 256 / CPR;                   // token rows per iteration
#pragma unroll
    for (int it = 0; it < TT / RPI; it++) {
        const int t = it * RPI + i / CPR, ch = i % CPR;
        us8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + t < Tp) v = *(const us8*)(in + ((int64_t)b * Tp + t0 + t) * ld_in + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) tile[ch * 8 + e][t] = v[e];
    }
    __syncthreads();
    constexpr int TPR = TT / 8;                      // 16-byte chunks per output row of the tile
    constexpr int CPI = 256 / TPR;                   // output rows per iteration
#pragma unroll
    for (int it = 0; it < TC / CPI; it++) {
        const int c = it * CPI + i / TPR, tch = i % TPR;
        if (t0 + tch * 8 < Tp) {                     // Tp % 8 == 0: a chunk is entirely inside or outside
            const unsigned* src = (const unsigned*)&tile[c][tch * 8];
            const uint4 o = make_uint4(src[0], src[1], src[2], src[3]);
            *(uint4*)(out + ((int64_t)b * ncols + c0 + c) * Tp + t0 + tch * 8) = o;
        }
    }
}

extern "C" int owl_transpose_tokens_bf16(void* stream, const void* in, int64_t ld_in, void* out, int64_t B, int64_t Tp, int64_t ncols,
                                         int64_t out_cols) {
    OWL_CHECK_ARG(in && out && B > 0 && Tp > 0 && ncols > 0, "owl_transpose_tokens_bf16: bad args");
    OWL_CHECK_ARG(Tp % 8 == 0 && ncols % 64 == 0 && ld_in % 8 == 0, "owl_transpose_tokens_bf16: Tp %% 8, ncols %% 64, ld_in %% 8");
    if (out_cols <= 0) out_cols = ncols;
    OWL_CHECK_ARG(out_cols >= ncols, "owl_transpose_tokens_bf16: out_cols < ncols");
    if (ncols % 128 == 0 && Tp >= 512) {             // 256-byte segments on both sides
        dim3 grid((unsigned)(ncols / 128), (unsigned)((Tp + 127) / 128), (unsigned)B);
        hipLaunchKernelGGL((transpose_tokens_kernel<128, 128>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, (int)Tp, (int)out_cols);
    } else {
        dim3 grid((unsigned)(ncols / 64), (unsigned)((Tp + 63) / 64), (unsigned)B);
        hipLaunchKernelGGL((transpose_tokens_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, (int)Tp, (int)out_cols);
    }
    OWL_LAUNCH_CHECK();
    return 0;
}

extern "C" int owl_transpose_bf16(void* stream, const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C) {
    OWL_CHECK_ARG(in && out && R > 0 && C > 0, "owl_transpose_bf16: bad args");
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C);
    OWL_LAUNCH_CHECK();
    return 0;
}
