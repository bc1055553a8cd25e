// Device input pipeline: ref src/dataset.py:69-71 = HF OwlViTImageProcessor (PIL backend):
// RGB u8 HWC --PIL bicubic--> SxS u8 --x(1/255), (x-mean)/std--> pixel_values [3,S,S].
//
// Pillow's resize is two separable fixed-point passes with a u8 round/clamp between them (Resample.c
// ImagingResampleHorizontal_8bpc / Vertical_8bpc); the tap tables come from owl_bicubic_coeffs (host, f64 like Pillow).
// Everything after the resize is a function of one u8 level per channel, so rescale+normalize is a [3][256] f32 table
// built by the caller with the reference's own float arithmetic -> the whole pipeline is bit-exact.  HBM-bound byte
// work (no MFMA): per image it reads H*W*3 B, writes/reads the H x S u8 intermediate, writes 3*S*S*(4|2) B.
#include "common.h"

#define PIL_PRECISION_BITS 22

__device__ __forceinline__ unsigned char pil_clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return (unsigned char)min(255, max(0, v));
}

// ---- one launch pair for a ragged batch (a per-image form was launch-bound -- 2 launches per image, 0.44 against 0.22 ms for 32 images -- and had no caller: removed in round 4) ----
// desc[i] = {src, H, W, bounds_x, kk_x, ksize_x, bounds_y, kk_y, ksize_y, tmp byte offset} as 10 int64 (device memory)
struct PreDesc { const unsigned char* src; long long H, W; const int* bx; const int* kx; long long ksx; const int* by; const int* ky; long long ksy; long long tmp_off; };

__global__ __launch_bounds__(256) void pp_resize_h_batch_kernel(const PreDesc* __restrict__ desc, unsigned char* __restrict__ tmp_base, int out_w) {
    const PreDesc d = desc[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w || y >= d.H) return;
    const int xmin = d.bx[2 * x], n = d.bx[2 * x + 1];
    const int* k = d.kx + (int64_t)x * d.ksx;
    const unsigned char* row = d.src + ((int64_t)y * d.W + xmin) * 3;
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; t++) {
        const int w = k[t];
        s0 += (int)row[3 * t] * w;
        s1 += (int)row[3 * t + 1] * w;
        s2 += (int)row[3 * t + 2] * w;
    }
    unsigned char* o = tmp_base + d.tmp_off + ((int64_t)y * out_w + x) * 3;
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
}

template <bool BF16>
__global__ __launch_bounds__(256) void pp_resize_v_batch_kernel(const PreDesc* __restrict__ desc, const unsigned char* __restrict__ tmp_base,
                                                                void* __restrict__ out, const float* __restrict__ lut, int out_h, int out_w) {
    const PreDesc d = desc[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w) return;
    const int ymin = d.by[2 * y], n = d.by[2 * y + 1];
    const int* k = d.ky + (int64_t)y * d.ksy;
    const unsigned char* col = tmp_base + d.tmp_off + ((int64_t)ymin * out_w + x) * 3;
    const int64_t stride = (int64_t)out_w * 3;
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; t++) {
        const int w = k[t];
        s0 += (int)col[t * stride] * w;
        s1 += (int)col[t * stride + 1] * w;
        s2 += (int)col[t * stride + 2] * w;
    }
    const float v0 = lut[pil_clip8(s0)], v1 = lut[256 + pil_clip8(s1)], v2 = lut[512 + pil_clip8(s2)];
    const int64_t plane = (int64_t)out_h * out_w, o = (int64_t)blockIdx.z * 3 * plane + (int64_t)y * out_w + x;
    if (BF16) {
        bf16_t* p = (bf16_t*)out;
        p[o] = f2bf(v0); p[plane + o] = f2bf(v1); p[2 * plane + o] = f2bf(v2);
    } else {
        float* p = (float*)out;
        p[o] = v0; p[plane + o] = v1; p[2 * plane + o] = v2;
    }
}

OWL_API int owl_preprocess_u8_batch(void* stream, const void* desc, int64_t n_images, int64_t max_h, unsigned char* tmp, const float* lut,
                                       void* out, int out_bf16, int64_t out_h, int64_t out_w) {
    OWL_CHECK_ARG(desc && tmp && lut && out, "owl_preprocess_u8_batch: null pointer");
    OWL_CHECK_ARG(n_images > 0 && n_images < 65536 && max_h > 0 && max_h < 65536 && out_h > 0 && out_h < 65536 && out_w > 0,
                  "owl_preprocess_u8_batch: bad sizes n=%lld max_h=%lld out=%lldx%lld", (long long)n_images, (long long)max_h, (long long)out_h, (long long)out_w);
    hipStream_t s = (hipStream_t)stream;
    const unsigned gx = (unsigned)((out_w + 255) / 256);
    hipLaunchKernelGGL(pp_resize_h_batch_kernel, dim3(gx, (unsigned)max_h, (unsigned)n_images), dim3(256), 0, s, (const PreDesc*)desc, tmp, (int)out_w);
    OWL_LAUNCH_CHECK();
    if (out_bf16)
        hipLaunchKernelGGL((pp_resize_v_batch_kernel<true>), dim3(gx, (unsigned)out_h, (unsigned)n_images), dim3(256), 0, s, (const PreDesc*)desc, tmp, out, lut, (int)out_h, (int)out_w);
    else
        hipLaunchKernelGGL((pp_resize_v_batch_kernel<false>), dim3(gx, (unsigned)out_h, (unsigned)n_images), dim3(256), 0, s, (const PreDesc*)desc, tmp, out, lut, (int)out_h, (int)out_w);
    OWL_LAUNCH_CHECK();
    return 0;
}

// ---- images that already have the model's size (a dataset that resizes on the host, or synthetic u8 pixels): only the table step is left ----
// src u8 [B,H,W,3] (HWC, what PIL / a DataLoader of raw images hands over) or [B,3,H,W] (CHW) -> out [B,3,H,W] f32 | bf16 = lut[c][level].
// Pillow's resize to the size an image already has returns a copy (Image.resize: `if self.size == size and box == (0, 0) + self.size: return self.copy()`),
// and its bicubic taps at scale 1 are (0, 1, 0, 0): this IS the reference pipeline for such an image, bit for bit.  Byte work, HBM-bound: 3 B read + 6 | 12 B
// written per pixel; a thread takes 4 pixels (HWC: three aligned 4-byte loads; CHW: one per plane) and writes 8 | 16 contiguous bytes per plane.
template <bool BF16, bool CHW>
__global__ __launch_bounds__(256) void pp_normalize_u8_kernel(const unsigned char* __restrict__ src, const float* __restrict__ lut, void* __restrict__ out,
                                                              int64_t n_img, int64_t plane) {
    __shared__ float s_lut[768];
    for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const int64_t quads = plane / 4;                           // per image
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_img * quads) return;
    const int64_t b = t / quads, q = t - b * quads;
    unsigned char v[3][4];
    if (CHW) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned w = *(const unsigned*)(src + (b * 3 + c) * plane + q * 4);
#pragma unroll
            for (int j = 0; j < 4; j++) v[c][j] = (unsigned char)(w >> (8 * j));
        }
    } else {
        const unsigned* p = (const unsigned*)(src + (b * plane + q * 4) * 3);
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        unsigned char by[12];
#pragma unroll
        for (int j = 0; j < 4; j++) { by[j] = (unsigned char)(w0 >> (8 * j)); by[4 + j] = (unsigned char)(w1 >> (8 * j)); by[8 + j] = (unsigned char)(w2 >> (8 * j)); }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) v[c][j] = by[3 * j + c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int64_t o = (b * 3 + c) * plane + q * 4;
        const float f0 = s_lut[c * 256 + v[c][0]], f1 = s_lut[c * 256 + v[c][1]], f2 = s_lut[c * 256 + v[c][2]], f3 = s_lut[c * 256 + v[c][3]];
        if (BF16) {
            uint2 pk;
            pk.x = (unsigned)f2bf(f0) | ((unsigned)f2bf(f1) << 16);
            pk.y = (unsigned)f2bf(f2) | ((unsigned)f2bf(f3) << 16);
            *(uint2*)((bf16_t*)out + o) = pk;
        } else {
            *(float4*)((float*)out + o) = make_float4(f0, f1, f2, f3);
        }
    }
}

OWL_API int owl_normalize_u8(void* stream, const unsigned char* src, int src_chw, const float* lut, void* out, int out_bf16, int64_t n_images, int64_t H, int64_t W) {
    OWL_CHECK_ARG(src && lut && out, "owl_normalize_u8: null pointer");
    OWL_CHECK_ARG(n_images > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "owl_normalize_u8: bad sizes n=%lld %lldx%lld (H*W %% 4 == 0)", (long long)n_images, (long long)H, (long long)W);
    OWL_CHECK_ARG(((uintptr_t)src & 3) == 0 && ((uintptr_t)out & 15) == 0, "owl_normalize_u8: src must be 4-byte, out 16-byte aligned");
    const int64_t plane = H * W, threads = n_images * (plane / 4);
    OWL_CHECK_ARG((threads + 255) / 256 < (1LL << 31), "owl_normalize_u8: batch too large for one launch");
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (out_bf16) {
        if (src_chw) hipLaunchKernelGGL((pp_normalize_u8_kernel<true, true>), grid, block, 0, s, src, lut, out, n_images, plane);
        else hipLaunchKernelGGL((pp_normalize_u8_kernel<true, false>), grid, block, 0, s, src, lut, out, n_images, plane);
    } else {
        if (src_chw) hipLaunchKernelGGL((pp_normalize_u8_kernel<false, true>), grid, block, 0, s, src, lut, out, n_images, plane);
        else hipLaunchKernelGGL((pp_normalize_u8_kernel<false, false>), grid, block, 0, s, src, lut, out, n_images, plane);
    }
    OWL_LAUNCH_CHECK();
    return 0;
}
