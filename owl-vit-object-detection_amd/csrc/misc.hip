// Small HBM-bound utilities: f32 -> bf16 cast, bf16 2-D transpose, fills.
#include "common.h"

// A thread converts four 8-element pieces per trip, every load of the trip issued before the first conversion (round 6: one piece per trip left a thread with 32
// bytes in flight and each trip behind the previous trip's store acknowledgement -- 4.1 TB/s; the f32 source is dead once read: streaming loads).
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t trip = nthreads * 32;                  // elements per trip of the whole grid: four pieces of nthreads * 8
    int64_t base = 0;
    for (; base + trip <= n; base += trip) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = base + ((int64_t)u * nthreads + tid) * 8;
            a[u] = ld_stream_f4(in + i); b[u] = ld_stream_f4(in + i + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = base + ((int64_t)u * nthreads + tid) * 8;
            uint4 o;
            o.x = pack_bf2(a[u].x, a[u].y); o.y = pack_bf2(a[u].z, a[u].w); o.z = pack_bf2(b[u].x, b[u].y); o.w = pack_bf2(b[u].z, b[u].w);
            *(uint4*)(out + i) = o;
        }
    }
    for (int64_t i = base + tid * 8; i < n; i += nthreads * 8) {
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

OWL_API int owl_cast_f32_bf16(void* stream, const float* in, void* out, int64_t n) {
    OWL_CHECK_ARG(in && out && n >= 0, "owl_cast_f32_bf16: null pointer");
    OWL_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "owl_cast_f32_bf16: pointers must be 16-byte aligned");
    if (n == 0) return 0;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;                    // (eight workgroups per CU: 27 trips of 32 elements per thread at the headline image batch)
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

OWL_API int owl_transpose_bf16(void* stream, const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C) {
    OWL_CHECK_ARG(in && out && R > 0 && C > 0, "owl_transpose_bf16: bad args");
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C);
    OWL_LAUNCH_CHECK();
    return 0;
}
