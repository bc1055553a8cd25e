// Fused AdamW over the flat trainable bucket (ref main.py:56-60,91: torch.optim.AdamW, ONE param group,
// lr 3e-6, wd 0.1 applied to every trainable tensor).  HBM-bound: 4 f32 streams in, 3 out (+ the bf16
// compute copy the next forward reads), 16-byte vector accesses.
#include "common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p_bf16, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float4 pv = *(const float4*)(p + i), gv = *(const float4*)(g + i), mv = *(const float4*)(m + i), vv = *(const float4*)(v + i);
        float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float gg = ga[e] * grad_scale;
            pa[e] *= (1.f - lr * wd);                       // decoupled weight decay
            ma[e] = b1 * ma[e] + (1.f - b1) * gg;
            va[e] = b2 * va[e] + (1.f - b2) * gg * gg;
            const float denom = sqrtf(va[e]) / bc2_sqrt + eps;
            pa[e] -= (lr / bc1) * (ma[e] / denom);
        }
        *(float4*)(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
        if (p_bf16) {
            uint2 o; o.x = pack_bf2(pa[0], pa[1]); o.y = pack_bf2(pa[2], pa[3]);
            *(uint2*)(p_bf16 + i) = o;
        }
    }
}

OWL_API int owl_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale) {
    OWL_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "owl_adamw_step: bad args (n %% 4 == 0, step >= 1)");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    OWL_LAUNCH_CHECK();
    return 0;
}
