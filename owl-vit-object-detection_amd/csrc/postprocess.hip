// Inference post-process on device: ref src/models.py:122-146 (PostProcess.__call__)
// + the eval loop's top-k (ref main.py:114-117).  Compiled with -ffp-contract=off so the IoU compare sees the same
// IEEE f32 results as the CPU arithmetic it replaces (torchvision.ops.batched_nms, absent from this stack).
//
//   per image:  score_p = max_c sims[p][c], class_p = first arg-max        (ref models.py:132-134)
//               keep p with score_p > confidence_threshold                 (ref models.py:136-139)
//               class-aware NMS, result ordered by descending score        (ref models.py:141-144)
//               -- torchvision.ops.batched_nms has TWO routes (torchvision/ops/boxes.py) and both are here, selected per call by `route`:
//                  0 per class on the raw coordinates (_batched_nms_vanilla); 1 coordinate offset: boxes + class * (max coordinate + 1) in
//                  f32, then class-agnostic NMS (_batched_nms_coordinate_trick); 2 / 3 = what torchvision itself picks for a tensor on a
//                  GPU / on the CPU (the coordinate trick up to 20 000 / 4 000 box coordinates past the threshold, per class above).
//                  The two differ only where the rounding of the shifted coordinates moves an IoU across the threshold.
//               optional prefix of max_out (= topk of an already sorted list; ref main.py:114-117)
//
// Three launches, no host sync, all HBM/latency-bound integer-and-compare work (no MFMA):
//   pp_sort_kernel   one 1024-thread workgroup per image: row max + threshold, 64-bit keys
//                    (score desc | patch asc | class) bitonic-sorted in LDS, sorted boxes/scores/classes written out
//   pp_mask_kernel   one wave per 64x64 block of the (sorted) pair matrix: bit t of word [i][cb] = "box i suppresses
//                    box cb*64+t" (same class, IoU > thr, lower score); column boxes broadcast with v_readlane
//   pp_scan_kernel   one wave per image walks the sorted list keeping a 64-bit-per-lane "removed" set; mask rows are
//                    fetched 8 ahead (they do not depend on the decisions), kept boxes are emitted in order
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned orderable(float f) {   // monotone map f32 -> u32
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

template <int NP>
__global__ __launch_bounds__(1024) void pp_sort_kernel(const float* __restrict__ sims, const float* __restrict__ boxes,
                                                       float* __restrict__ s_box, float* __restrict__ s_score,
                                                       int* __restrict__ s_cls, int* __restrict__ s_idx,
                                                       int* __restrict__ n_valid, float* __restrict__ max_coord, int P, int C, float conf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_max;
    u64* keys = (u64*)smem;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_max = 0u;                        // (orderable(-inf-like): any real coordinate maps above 0)
    for (int p = tid; p < NP; p += 1024) {
        u64 key = ~0ull;
        if (p < P) {
            const float* r = sims + ((int64_t)b * P + p) * C;
            float best = r[0];
            int arg = 0;
            for (int c = 1; c < C; c++) {
                const float v = r[c];
                if (v > best) { best = v; arg = c; }
            }
            if (best > conf) key = ((u64)(~orderable(best)) << 32) | ((u64)(unsigned)p << 16) | (u64)(unsigned)arg;
        }
        keys[p] = key;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NP / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 a = keys[i], c = keys[l];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { keys[i] = c; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    if (tid == 0 && keys[0] == ~0ull) n_valid[b] = 0;
    unsigned lmax = 0u;                              // `boxes.max()` over the boxes past the threshold (coordinate-offset route)
    for (int p = tid; p < P; p += 1024) {
        const u64 key = keys[p];
        if (key == ~0ull) continue;
        const int idx = (int)((key >> 16) & 0xFFFF), cls = (int)(key & 0xFFFF);
        const int64_t o = (int64_t)b * P + p;
        const float4 bx = ((const float4*)boxes)[(int64_t)b * P + idx];
        ((float4*)s_box)[o] = bx;
        lmax = max(max(lmax, orderable(bx.x)), max(max(orderable(bx.y), orderable(bx.z)), orderable(bx.w)));
        s_score[o] = unorderable(~(unsigned)(key >> 32));
        s_cls[o] = cls;
        s_idx[o] = idx;
        if (p + 1 == P || keys[p + 1] == ~0ull) n_valid[b] = p + 1;
    }
    atomicMax(&s_max, lmax);
    __syncthreads();
    if (tid == 0) max_coord[b] = s_max ? unorderable(s_max) : 0.f;
}

__global__ __launch_bounds__(64) void pp_mask_kernel(const float* __restrict__ s_box, const int* __restrict__ s_cls,
                                                     const int* __restrict__ n_valid, const float* __restrict__ max_coord,
                                                     u64* __restrict__ mask, int P, int W, float thr, int route) {
    const int cb = blockIdx.x, rb = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int n = n_valid[b];
    // torchvision's own dispatch: boxes.numel() (= 4 n) > 20 000 on a GPU / > 4 000 on the CPU -> per class, else the coordinate trick
    const bool shifted = route == 1 || (route == 2 && 4 * n <= 20000) || (route == 3 && 4 * n <= 4000);
    const int i = rb * 64 + lane;
    if (rb * 64 >= n) return;                       // rows >= n_valid are never read by the scan
    u64 bits = 0;
    if (cb >= rb && cb * 64 < n) {
        const int64_t base = (int64_t)b * P;
        const int ii = min(i, P - 1), jj = min(cb * 64 + lane, P - 1);
        float4 bi = ((const float4*)s_box)[base + ii];
        int ci = s_cls[base + ii];
        float4 bj = ((const float4*)s_box)[base + jj];
        int cj = s_cls[base + jj];
        if (shifted) {          // boxes + class * (max + 1), f32 (compiled -ffp-contract=off: a multiply and an add, as torch does them)
            const float step = max_coord[b] + 1.0f;
            const float oi = (float)ci * step, oj = (float)cj * step;
            bi.x += oi; bi.y += oi; bi.z += oi; bi.w += oi;
            bj.x += oj; bj.y += oj; bj.z += oj; bj.w += oj;
            ci = cj = 0;        // one class-agnostic NMS over the shifted boxes
        }
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        const float aj = (bj.z - bj.x) * (bj.w - bj.y);
#pragma unroll
        for (int t = 0; t < 64; t++) {
            const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bj.x), t));
            const float y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bj.y), t));
            const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bj.z), t));
            const float y2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bj.w), t));
            const float at = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aj), t));
            const int ct = __builtin_amdgcn_readlane(cj, t);
            const int j = cb * 64 + t;
            const float w = fmaxf(0.f, fminf(bi.z, x2) - fmaxf(bi.x, x1));
            const float h = fmaxf(0.f, fminf(bi.w, y2) - fmaxf(bi.y, y1));
            const float inter = w * h;
            const float ovr = inter / (ai + at - inter);
            if (j > i && j < n && ct == ci && ovr > thr) bits |= 1ull << t;
        }
    }
    if (i < n) mask[((int64_t)b * P + i) * W + cb] = bits;
}

__global__ __launch_bounds__(64) void pp_scan_kernel(const u64* __restrict__ mask, const float* __restrict__ s_box,
                                                     const float* __restrict__ s_score, const int* __restrict__ s_cls,
                                                     const int* __restrict__ s_idx, const int* __restrict__ n_valid,
                                                     float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                     int64_t* __restrict__ out_classes, int64_t* __restrict__ out_patch,
                                                     int* __restrict__ out_count, int P, int W, int max_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = n_valid[b];
    const u64* m = mask + (int64_t)b * P * W;
    const int64_t base = (int64_t)b * P;
    u64 remv0 = 0, remv1 = 0;
    int kept = 0;
    for (int i0 = 0; i0 < n && kept < max_out; i0 += 8) {
        u64 r0[8], r1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int row = min(i0 + u, n - 1);
            r0[u] = lane < W ? m[(int64_t)row * W + lane] : 0ull;
            r1[u] = lane + 64 < W ? m[(int64_t)row * W + lane + 64] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u;
            if (i < n && kept < max_out) {
                const int w = i >> 6;
                const u64 src = w < 64 ? remv0 : remv1;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)src, w & 63);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(src >> 32), w & 63);
                const u64 word = ((u64)hi << 32) | lo;
                if (!((word >> (i & 63)) & 1ull)) {
                    if (lane == 0) {
                        const int64_t o = (int64_t)b * max_out + kept;
                        ((float4*)out_boxes)[o] = ((const float4*)s_box)[base + i];
                        out_scores[o] = s_score[base + i];
                        out_classes[o] = s_cls[base + i];
                        out_patch[o] = s_idx[base + i];
                    }
                    kept++;
                    remv0 |= r0[u];
                    remv1 |= r1[u];
                }
            }
        }
    }
    if (lane == 0) out_count[b] = kept;
}

static int64_t pp_ws_bytes(int64_t B, int64_t P) {
    const int64_t W = (P + 63) / 64;
    return B * P * (16 + 4 + 4 + 4) + 256 + B * P * W * 8 + B * 4 + B * 4 + 256;
}

OWL_API int owl_postprocess_workspace(int64_t B, int64_t P, int64_t* bytes) {
    OWL_CHECK_ARG(B > 0 && P > 0 && P <= 8192 && bytes, "owl_postprocess_workspace: need B > 0, 0 < P <= 8192 (B=%lld P=%lld)", (long long)B, (long long)P);
    *bytes = pp_ws_bytes(B, P);
    return 0;
}

OWL_API int owl_postprocess(void* stream, const float* boxes, const float* sims, void* workspace, int64_t ws_bytes,
                               float* out_boxes, float* out_scores, int64_t* out_classes, int64_t* out_patch, int* out_count,
                               int64_t B, int64_t P, int64_t C, int64_t max_out, float conf_thr, float iou_thr, int route) {
    OWL_CHECK_ARG(boxes && sims && workspace && out_boxes && out_scores && out_classes && out_patch && out_count, "owl_postprocess: null pointer");
    OWL_CHECK_ARG(B > 0 && P > 0 && P <= 8192 && C > 0 && C < 65536 && max_out > 0, "owl_postprocess: need 0 < P <= 8192, 0 < C < 65536, max_out > 0 (P=%lld C=%lld max_out=%lld)", (long long)P, (long long)C, (long long)max_out);
    OWL_CHECK_ARG(route >= 0 && route <= 3, "owl_postprocess: route must be 0 (per class), 1 (coordinate offset), 2 (torchvision's choice on a GPU) or 3 (... on the CPU)");
    OWL_CHECK_ARG(ws_bytes >= pp_ws_bytes(B, P), "owl_postprocess: workspace %lld < required %lld bytes", (long long)ws_bytes, (long long)pp_ws_bytes(B, P));
    OWL_CHECK_ARG(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)boxes & 15) == 0 && ((uintptr_t)out_boxes & 15) == 0, "owl_postprocess: boxes / out_boxes / workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int W = (int)((P + 63) / 64);
    unsigned char* ws = (unsigned char*)workspace;
    float* s_box = (float*)ws;                 ws += B * P * 16;
    float* s_score = (float*)ws;               ws += B * P * 4;
    int* s_cls = (int*)ws;                     ws += B * P * 4;
    int* s_idx = (int*)ws;                     ws += B * P * 4;
    ws = (unsigned char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    u64* mask = (u64*)ws;                      ws += B * P * (int64_t)W * 8;
    int* n_valid = (int*)ws;                   ws += B * 4;
    float* max_coord = (float*)ws;
    if (P <= 1024) hipLaunchKernelGGL((pp_sort_kernel<1024>), dim3((unsigned)B), dim3(1024), 1024 * 8, s, sims, boxes, s_box, s_score, s_cls, s_idx, n_valid, max_coord, (int)P, (int)C, conf_thr);
    else if (P <= 4096) hipLaunchKernelGGL((pp_sort_kernel<4096>), dim3((unsigned)B), dim3(1024), 4096 * 8, s, sims, boxes, s_box, s_score, s_cls, s_idx, n_valid, max_coord, (int)P, (int)C, conf_thr);
    else hipLaunchKernelGGL((pp_sort_kernel<8192>), dim3((unsigned)B), dim3(1024), 8192 * 8, s, sims, boxes, s_box, s_score, s_cls, s_idx, n_valid, max_coord, (int)P, (int)C, conf_thr);
    OWL_LAUNCH_CHECK();
    hipLaunchKernelGGL(pp_mask_kernel, dim3(W, W, (unsigned)B), dim3(64), 0, s, s_box, s_cls, n_valid, max_coord, mask, (int)P, W, iou_thr, route);
    OWL_LAUNCH_CHECK();
    hipLaunchKernelGGL(pp_scan_kernel, dim3((unsigned)B), dim3(64), 0, s, mask, s_box, s_score, s_cls, s_idx, n_valid, out_boxes, out_scores, out_classes, out_patch, out_count, (int)P, W, (int)max_out);
    OWL_LAUNCH_CHECK();
    return 0;
}
